"""Parse '[pq wg] job tile start_us end_us xcc hwid lds' lines (PQ_SUITE_DEBUG=2) of the LAST step in a log:
residency per CU over time, LDS in use, per-WG durations."""
import sys, collections
rows = [l.split()[2:] for l in open(sys.argv[1]) if l.startswith("[pq wg]")]
njobs = len({r[0] for r in rows})
# keep the last step: lines repeat per run; take the last block of equal size
per_step = collections.OrderedDict()
for r in rows: per_step.setdefault((r[0], r[1]), []).append(r)
last = [v[-1] for v in per_step.values()]
ev = []
cu_of = {}
for j, x, s, e, xcc, hw, lds in last:
    s, e, xcc, hw, lds = float(s), float(e), int(xcc) & 0xf, int(hw), int(lds)
    cu = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)  # xcc, se, sh, cu
    ev.append((s, 1, lds, cu)); ev.append((e, -1, -lds, cu))
ev.sort()
t_end = max(e[0] for e in ev)
print(f"{len(last)} workgroups, {len({e[3] for e in ev})} distinct CUs, span {t_end:.0f} us")
# time-average residency
n = 0; lds = 0; tprev = 0.0; acc_n = 0.0; acc_l = 0.0
buckets = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
for t, dn, dl, cu in ev:
    dt = t - tprev
    acc_n += n * dt; acc_l += lds * dt
    b = int(tprev // 1000); buckets[b][0] += n * dt; buckets[b][1] += lds * dt; buckets[b][2] += dt
    n += dn; lds += dl; tprev = t
ncu = len({e[3] for e in ev})
print(f"average resident workgroups per CU: {acc_n / t_end / ncu:.2f}; average LDS in use per CU: {acc_l / t_end / ncu / 1024:.1f} KB")
for b in sorted(buckets):
    a = buckets[b]
    if a[2] > 0: print(f"  t={b}..{b+1} ms: {a[0]/a[2]/ncu:5.2f} WG/CU  {a[1]/a[2]/ncu/1024:6.1f} KB LDS/CU")
# start-time distribution per LDS class
cls = collections.defaultdict(list)
for j, x, s, e, xcc, hw, lds in last:
    l = int(lds); c = "<=14K" if l and l <= 14336 else "<=28K" if l and l <= 28672 else ">28K" if l else "gather"
    cls[c].append((float(s), float(e)))
for c, v in cls.items():
    st = sorted(s for s, e in v); du = sorted(e - s for s, e in v)
    q = lambda a, f: a[int(f * (len(a) - 1))]
    print(f"  class {c:6s}: {len(v):4d} WGs  start p10/p50/p90/max = {q(st,.1):6.0f} {q(st,.5):6.0f} {q(st,.9):6.0f} {st[-1]:6.0f} us   duration p10/p50/p90 = {q(du,.1):6.0f} {q(du,.5):6.0f} {q(du,.9):6.0f} us")
