"""Print the kernel timeline of one replayed iteration from a rocprofv3 --kernel-trace CSV (developer tool)."""
import csv, sys
path = sys.argv[1]
full = len(sys.argv) > 2
rows = list(csv.DictReader(open(path)))
def nm(r):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    return n.split('(')[0][:60]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), nm(r), r.get('Queue_Id', '')) for r in rows)
# anchor: a kernel that runs once per iteration (the loops differ: hidden<-visual backward, else the blend backward)
vb = []
for anchor in ('visual_backward', 'blend_backward_lanes_kernel<3, 3', 'blend_backward_lanes_kernel<1, 3', 'blend_backward_lanes_kernel<3, 2', 'blend_backward_lanes_kernel', 'blend_backward_kernel<3, 3', 'blend_backward_kernel<1, 3', 'blend_backward_kernel<3, 1',
               'blend_backward_kernel<1, 1', 'blend_backward_kernel<3, 2',
               'blend_backward_kernel<3, 0'):
    vb = [e for e in ev if anchor in e[2]]
    if len(vb) > 2:
        break
gaps = [(vb[i + 1][0] - vb[i][0]) / 1e6 for i in range(len(vb) - 1)]
if len(sys.argv) > 3 and sys.argv[3] == "spacings":  # every iteration's length, in trace order (graph replays: k per replay)
    print('iteration spacings (us):', ' '.join('%.0f' % (g * 1e3) for g in gaps))
# the replayed iterations are the bulk of the anchors: take the window with the MEDIAN spacing (the shortest one may
# belong to the rasteriser-only timing the bench runs afterwards, the longest to an eager or capturing iteration)
i = sorted(range(len(gaps)), key=lambda k: gaps[k])[len(gaps) // 2]
w0, w1 = vb[i][1], vb[i + 1][1]
win = [e for e in ev if e[0] >= w0 and e[1] <= w1]
busy, last = 0, w0
for s, e, n, q in win:
    if e > last:
        busy += e - max(s, last)
        last = e
print('window ms', (w1 - w0) / 1e6, 'kernels', len(win), 'sum', sum(e[1] - e[0] for e in win) / 1e6, 'busy', busy / 1e6)
agg = {}
for s, e, n, q in win:
    k = n if ('fnx' in n or 'kernel' in n and 'at::' not in n) else 'torch/other'
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%8.1f us %4d  %s' % (t / 1e3, c, k))
if full:
    prev = w0
    for s, e, n, q in win:
        print('%7.1f +%5.1f dur %6.1f q%s %s' % ((s - w0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, q, n))
        prev = e
