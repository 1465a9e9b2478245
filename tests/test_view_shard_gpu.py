"""View-group sharding on the real kernels: two processes share the one GPU of the test box, each holds three of the six views
(`engine.ViewShard` over a gloo group — device tensors are staged through host memory, RCCL refuses two ranks on one device).
What this adds to the gloo/emulation test (tests/test_parallel_gloo.py): the HIP convs over the widened bands (widths 50 / 26 /
14, stride 2 with the two-column left halo, the folded nearest-x2), e4m3 lo planes travelling as halos, the GroupNorm apply
kernel on combined records, and `pnc_attn_views` with kv_views = n_local + 2."""
import os
import sys
import tempfile
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

from helpers import measured

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
T = 2


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from helpers import cond as cond_of, product_network, step_inputs
    from panacea_amd import configs, hip, parallel
    hip.load()
    parallel.init_distributed("gloo")
    lo = parallel.RankLayout(world, rank, views=world)
    groups = parallel.Groups(lo)
    vs = groups.view_shard()
    kw = configs.with_frames(configs.get("tiny"), T)
    net, _, _ = product_network("tiny", kw=kw, device="cuda")
    inp = step_inputs("tiny", kw, device="cuda", t_index=500, shape=(2, T, 8, 96))
    loc = {k: (parallel.local_views(v, lo) if v.dim() == 4 else v) for k, v in inp.items()}
    parallel.apply_view_shard(net, vs)
    with torch.no_grad():
        eps = net(loc["x"], loc["t"], cond_of(loc))
        torch.cuda.synchronize()
        full = parallel.gather_views(eps, groups)
        if rank == 0:
            parallel.apply_view_shard(net, None)
            ref = net(inp["x"], inp["t"], cond_of(inp))
            torch.save({"sharded": full.cpu(), "single": ref.cpu(), "exchanges": vs.exchanges, "bytes": vs.bytes_sent,
                        "precision": net.diffusion_model.precision}, Path(out_dir) / "out.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_view_bands_on_the_kernels_reproduce_the_single_process_eps():
    port = 29500 + ((os.getpid() * 13 + 5) % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        r = torch.load(Path(d) / "out.pt")
    diff = (r["sharded"] - r["single"]).abs()
    wl = diff.shape[-1] // 2
    edge = diff[..., [wl - 1, wl]].mean().item()            # the two columns either side of the band boundary
    print(f"view bands vs single process ({r['precision']}): max {diff.max().item():.3e} mean {diff.mean().item():.3e} "
          f"edge-mean {edge:.3e}; {r['exchanges']} exchanges, {r['bytes'] / 1e6:.2f} MB sent per rank")
    measured("view_shard_vs_single", max_abs=diff.max().item(), mean_abs=diff.mean().item(), edge_mean=edge,
             exchanges=r["exchanges"], ref_max=r["single"].abs().max().item())
    # Measured on MI355X (round 3): max 7.5e-4, mean 1.32e-4, edge-mean 1.22e-4 at |eps| <= 2.7.  Per output element the kernels
    # are width-invariant; the panorama statistics are combined from other partial records (1e-7 relative), which decorrelates
    # the fp16 operand roundings downstream: the two evaluations differ like two `precise` runs differ from the oracle.  A halo /
    # neighbour-view mistake is O(0.1) and sits at the band edge.
    assert diff.max().item() <= 1.2e-3 and diff.mean().item() <= 2.0e-4
    assert edge <= 3.0 * diff.mean().item() + 1e-6
    assert r["exchanges"] > 50
