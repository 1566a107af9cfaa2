"""Kernel-to-kernel view of one optimisation step from a -DAGS_TIMELINE -DAGS_TL_REALTIME build (stamps = the chip-wide
100 MHz counter, comparable across CUs): when each kernel's first wave starts and last wave ends (gaps between
kernels), the start-time distribution (launch ramp, generations of waves) and, for the per-Gaussian kernels, the phases.
    python profiles/experiments/build_exp.py "tlrt=-DAGS_TIMELINE,-DAGS_TL_REALTIME"
    AGS_TL_DUMP=gpurun_out/tlrt.npz AGS_LIB_PATH=scratch/libags_tlrt.so python profiles/experiments/timeline.py [--pipeline] [--graph 3]
    python profiles/experiments/timeline_realtime.py gpurun_out/tlrt.npz"""
import numpy as np, sys
t=np.load(sys.argv[1])['t']
NS=10.0
names={0:('preprocess/roleB',5),5:('tile_sort',1),2:('fwd',4),3:('bwd',5),4:('rows/roleA',1)}
g0=min(t[k][:,0][t[k][:,0]>0].min() for k in names if (t[k][:,0]>0).any())
for kid in (5,2,3,4,0):
    name,nph=names[kid]
    a=t[kid]; ok=a[:,0]>0; a=a[ok]
    if len(a)==0: continue
    end=np.where(a[:,1:nph+1]>0,a[:,1:nph+1],0).max(axis=1)
    s0=a[:,0].min(); e1=end.max()
    st=(a[:,0]-s0)*NS/1e3; life=(end-a[:,0])*NS/1e3
    print(f"{name:18s} waves {len(a):6d} first start @{(s0-g0)*NS/1e3:8.2f} last end @{(e1-g0)*NS/1e3:8.2f} span {(e1-s0)*NS/1e3:6.2f} | starts p50 {np.percentile(st,50):.2f} p90 {np.percentile(st,90):.2f} max {st.max():.2f} | life p50 {np.percentile(life,50):.2f} p90 {np.percentile(life,90):.2f} max {life.max():.2f}")
    if kid==0:
        ph=[(a[:,i+1]-a[:,i])*NS/1e3 for i in range(5)]
        print("   roleB phases p50:", [round(float(np.percentile(p,50)),2) for p in ph], "p90:", [round(float(np.percentile(p,90)),2) for p in ph])
    if kid==4:
        busy=life>2.0
        print("   roleA busy waves:", int(busy.sum()), "life p50 %.2f max %.2f"%(np.percentile(life[busy],50) if busy.any() else 0, life.max()))

# ---- phases of the member-row waves of the pipelined per-Gaussian kernel (stamps 2..5 exist only there)
a = t[4]; ok = (a[:, 0] > 0) & (a[:, 5] > 0); a = a[ok]
if len(a):
    print("member-row waves of the pipelined kernel:", len(a))
    for n, (i, j) in zip(["start -> chain rule done", "moments + Adam", "next view: projection", "next view: key emission", "tail"],
                         [(0, 2), (2, 3), (3, 4), (4, 5), (5, 1)]):
        d = (a[:, j] - a[:, i]) * 10 / 1e3
        print(f"  {n:26s} p50 {np.percentile(d, 50):6.2f} p90 {np.percentile(d, 90):6.2f} max {d.max():6.2f} us")
