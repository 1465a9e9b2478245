#!/bin/bash
# round 5: kernel trace of the default (two-stream) bench: how much of a step is the GPU idle or under-filled?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o
mkdir -p $O
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-baseline none --no-kernel-breakdown --no-modes --no-parity"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- $B > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/r5o/timeline.txt
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
ev=[(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id','?'), int(r.get('Workgroup_Size',0) or 0), int(r.get('Grid_Size',0) or 0)) for r in rows]
ev.sort()
# the last 3 steps: find cfg_euler_step kernels as step boundaries
marks=[e[1] for e in ev if 'cfg_euler_step' in e[2]]
print("step marks:", len(marks))
if len(marks) >= 3:
    t0, t1 = marks[-3], marks[-1]
    win=[e for e in ev if e[0] >= t0 and e[1] <= t1]
    span=(t1-t0)/1e6
    # union of busy intervals
    busy=0; cur_s=cur_e=None
    for s,e,*_ in sorted(win):
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e-cur_s
            cur_s, cur_e = s, e
        else: cur_e=max(cur_e,e)
    if cur_e is not None: busy += cur_e-cur_s
    print(f"2 steps: span {span:.2f} ms, some kernel in flight {busy/1e6:.2f} ms, idle {span-busy/1e6:.2f} ms ({(1-busy/1e6/span)*100:.1f} %)")
    # time with exactly one kernel in flight whose grid is small (< 256 workgroups)
    pts=[]
    for s,e,n,q,wg,grid in win:
        small = (grid // max(1,wg)) < 256
        pts.append((s,1,small)); pts.append((e,-1,small))
    pts.sort()
    n_all=n_small=0; last=pts[0][0]; t_one_small=0; t_two=0; t_one=0
    for t,d,small in pts:
        dt=t-last; last=t
        if n_all==1: t_one+=dt
        if n_all>=2: t_two+=dt
        if n_all>=1 and n_all==n_small: t_one_small+=dt
        n_all+=d; n_small+= d if small else 0
    print(f"one kernel in flight {t_one/1e6:.2f} ms, two or more {t_two/1e6:.2f} ms, only kernels of < 256 workgroups in flight {t_one_small/1e6:.2f} ms")
    queues={}
    for s,e,n,q,*_ in win: queues[q]=queues.get(q,0)+(e-s)
    print("kernel time per queue (ms):", {k: round(v/1e6,2) for k,v in queues.items()})
    # the largest gaps
    gaps=[]; cur_e=None
    for s,e,n,*_ in sorted(win):
        if cur_e is not None and s>cur_e: gaps.append((s-cur_e, n[:60]))
        cur_e = e if cur_e is None else max(cur_e,e)
    gaps.sort(reverse=True)
    print("gaps:", len(gaps), "sum", round(sum(g for g,_ in gaps)/1e6,3), "ms; largest:", [(round(g/1e3,1), n) for g,n in gaps[:8]])
PY
