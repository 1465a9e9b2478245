"""Full-size pin (build container only; ~15 min, ~30 GB): run the REFERENCE network of BASELINE
config 3 — Panacea+ stage-2, (B, T) = (2, 8), 32x384 latent, 256x3072 BEV hint — on the deterministic
synthetic weights/inputs and store a stride-7 sample of eps in tests/golden/full_cfg3.npz.  The CPU
oracle is run on the same data and must agree (fp32 vs fp32) before the file is written.

    python -m oracle.gen_golden_full                         # t = 999, input salt 0 -> full_cfg3.npz
    python -m oracle.gen_golden_full --t 500 --no-oracle     # further pins (round 3): full_cfg3_t500.npz
    python -m oracle.gen_golden_full --t 39 --salt 1 --no-oracle   # second input seed:  full_cfg3_t39_s1.npz
    python -m oracle.gen_golden_full --frames 1 --no-oracle        # BASELINE config 2 (round 4): full_cfg2.npz
    python -m oracle.gen_golden_full --yaml-exact --no-oracle      # BASELINE config 5, sampler step 0: full_cfg5_step0.npz
    python -m oracle.gen_golden_full --t 500 --wsalt 1 --no-oracle # second WEIGHT draw (round 5): full_cfg3_t500_w1.npz
    python -m oracle.gen_golden_full --t 500 --wtail 64 --no-oracle    # heavy-tailed stream (round 5): full_cfg3_t500_tail64.npz
    ... --full-eps                                                 # round 6: the file also holds the WHOLE eps (key "eps", fp32)

The extra pins skip the (6 minute) oracle leg: oracle vs reference is established by the first file and by the
small configurations; what the extra files pin is the reference's eps at other noise levels / inputs.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import panacea_oracle as po, ref_import          # noqa: E402
from oracle.gen_golden import GOLDEN, oracle_cfg              # noqa: E402
from panacea_amd import configs, synth                        # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--t", type=int, default=999, help="timestep index of every frame")
    ap.add_argument("--salt", type=int, default=0, help="salt of the synthetic INPUTS (weights keep salt 0)")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--frames", type=int, default=8, help="1 = BASELINE config 2: the YAML network built with num_frames = 1")
    ap.add_argument("--wsalt", type=int, default=0, help="salt of the synthetic WEIGHTS (synth.synth_state_dict)")
    ap.add_argument("--wtail", type=float, default=0.0, help="heavy-tail weight set: gain of the output channels c %% 64 == 5 of the residual-out tensors")
    ap.add_argument("--full-eps", action="store_true",
                    help="round 6 (SURVEY 8c full-size pin): store the WHOLE eps (fp32, 3.1 MB) next to the stride-7 sample")
    ap.add_argument("--yaml-exact", action="store_true",
                    help="BASELINE config 5: last-frame concat conditioning + share-noise latent, t = 999 (sampler step 0)")
    args = ap.parse_args()
    tag = "" if (args.t == 999 and args.salt == 0) else f"_t{args.t}" + (f"_s{args.salt}" if args.salt else "")
    if args.wsalt:
        tag += f"_w{args.wsalt}"
    if args.wtail:
        tag += f"_tail{int(args.wtail)}"
    stem = "full_cfg3"
    if args.frames != 8:
        assert args.frames == 1 and not args.yaml_exact
        stem = "full_cfg2"
    if args.yaml_exact:
        assert args.t == 999
        stem = "full_cfg5_step0"
    ns = ref_import.import_reference()
    kw = configs.with_frames(configs.get("full"), args.frames)
    t0 = time.time()
    net, wrapper = ref_import.build_reference_network(ns, kw)
    manifest = {k: list(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(manifest, salt=args.wsalt, tail=args.wtail)
    net.load_state_dict(sd, strict=True)
    del sd
    print(f"built + loaded in {time.time() - t0:.0f}s", flush=True)
    B, T, h, w = configs.SHAPES["full"]
    T = args.frames
    if args.yaml_exact:
        inp = synth.yaml_exact_step0_inputs(T, h, w, context_dim=kw["context_dim"], salt=args.salt)
    else:
        inp = synth.synth_inputs(B, T, h, w, context_dim=kw["context_dim"], t_index=args.t, salt=args.salt)
    c = {k: inp[k].clone() for k in ("concat", "crossattn", "cond_feat")}
    t0 = time.time()
    with torch.no_grad():
        eps = wrapper(inp["x"].clone(), inp["t"].clone(), c)
    t_ref = time.time() - t0
    print(f"reference forward {t_ref:.1f}s ({torch.get_num_threads()} threads); eps rms {eps.pow(2).mean().sqrt():.4f} "
          f"max {eps.abs().max():.4f}", flush=True)
    t_or = 0.0
    if not args.no_oracle:
        sd = {k: v.detach() for k, v in net.state_dict().items()}
        t0 = time.time()
        eps_o = po.wrapper_forward(sd, oracle_cfg(kw), inp["x"], inp["t"], {k: inp[k] for k in ("concat", "crossattn", "cond_feat")})
        t_or = time.time() - t0
        d = (eps - eps_o).abs().max().item()
        print(f"oracle forward {t_or:.1f}s; oracle vs reference max-abs {d:.3e}", flush=True)
        assert d <= 1e-4
    extra = {"eps": eps.numpy()} if args.full_eps else {}
    np.savez_compressed(GOLDEN / f"{stem}{tag}.npz", eps_s7=eps.reshape(-1)[::7].numpy(), **extra,
                        t_index=np.int32(args.t), input_salt=np.int32(args.salt),
                        weight_salt=np.int32(args.wsalt), weight_tail=np.float32(args.wtail),
                        eps_rms=np.float32(eps.pow(2).mean().sqrt().item()),
                        eps_max=np.float32(eps.abs().max().item()),
                        ref_seconds=np.float32(t_ref), oracle_seconds=np.float32(t_or),
                        threads=np.int32(torch.get_num_threads()))
    print("written", flush=True)
