"""Wave-slot residency of the lighting launch from the raw per-tile stamp records (tools/lighting_stamps.py with LV_STAMP_DUMP=<prefix>;
offline, no GPU).  Answers: how many waves of the launch are alive per SIMD over time, what separates two tiles that used the same
hardware wave slot, how long a workgroup's registers / LDS stay allocated after its first wave has finished.
    python tools/lighting_residency.py gpurun_out/<tag>/stamps_base_0.npz [waves_per_workgroup]
HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13; record word 3 = XCC_ID << 28 | HW_ID & 0x0fffffff."""
import sys
import numpy as np

path = sys.argv[1]
lw = int(sys.argv[2]) if len(sys.argv) > 2 else 4
both = np.load(path)['records']
idx = np.arange(len(both))
r = both[:, 0]
valid = (r[:, 0] != 0) | (r[:, 1] != 0)
r, idx = r[valid], idx[valid]
t0 = r[:, 0].astype(np.int64); t1 = r[:, 1].astype(np.int64)
base = t0.min(); t0 -= base; t1 -= base
hw = r[:, 3]
xcc = hw >> 28; wave = hw & 15; simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simd_key = cu_key * 4 + simd
slot_key = simd_key * 16 + wave
span = t1.max()
print('%s: %d tiles, span %.1f us, mean tile %.2f us' % (path, len(r), span * 0.01, (t1 - t0).mean() * 0.01))
print('distinct CUs %d, SIMDs %d, wave slots %d' % (len(np.unique(cu_key)), len(np.unique(simd_key)), len(np.unique(slot_key))))
nsimd = len(np.unique(simd_key))

# live waves per SIMD over time (10 ns ticks)
ticks = int(span) + 1
delta = np.zeros(ticks + 1, np.int64)
np.add.at(delta, t0, 1); np.add.at(delta, t1, -1)
live = np.cumsum(delta)[:ticks]
print('live waves per SIMD over the launch, by decile of its span: ' + ' '.join('%.2f' % (live[int(k * ticks / 10):int((k + 1) * ticks / 10)].mean() / nsimd) for k in range(10)))
print('mean live waves per SIMD: %.2f (first tenth and last tenth excluded: %.2f)' % (live.mean() / nsimd, live[ticks // 10: 9 * ticks // 10].mean() / nsimd))

# per SIMD: distribution of concurrently live waves in the steady part
order = np.argsort(simd_key, kind='stable')
lo, hi = ticks // 10, 9 * ticks // 10
hist = np.zeros(12)
for k in np.unique(simd_key)[::16]:
    m = simd_key == k
    d = np.zeros(ticks + 1, np.int64); np.add.at(d, t0[m], 1); np.add.at(d, t1[m], -1)
    l = np.cumsum(d)[lo:hi]
    hist += np.bincount(l, minlength=12)[:12]
print('share of time a SIMD holds n live waves (steady part, every 16th SIMD): ' + ' '.join('%d:%.3f' % (n, hist[n] / hist.sum()) for n in range(8)))

# gaps on one hardware wave slot: end of a tile -> start of the next tile in the same slot
o = np.lexsort((t0, slot_key))
sk, a0, a1 = slot_key[o], t0[o], t1[o]
same = sk[1:] == sk[:-1]
gap = (a0[1:] - a1[:-1])[same] * 0.01
print('gap between two tiles in one hardware wave slot (us): n %d  p5 %.2f  p25 %.2f  median %.2f  p75 %.2f  p95 %.2f  mean %.2f' %
      (len(gap), *np.percentile(gap, [5, 25, 50, 75, 95]), gap.mean()))

# workgroups: lw consecutive tile records
wg = idx // lw
o = np.argsort(wg, kind='stable')
uw, start = np.unique(wg[o], return_index=True)
cnt = np.diff(np.append(start, len(o)))
full = cnt == lw
st = np.minimum.reduceat(t0[o], start)[full]; en = np.maximum.reduceat(t1[o], start)[full]
first_end = np.minimum.reduceat(t1[o], start)[full]
wave_sum = np.add.reduceat((t1 - t0)[o], start)[full]
print('workgroups of %d waves: %d; life (first start -> last end) mean %.2f us; its waves busy %.2f of life x %d; first wave done %.2f us before the last' %
      (lw, full.sum(), (en - st).mean() * 0.01, (wave_sum / ((en - st) * lw)).mean(), lw, (en - first_end).mean() * 0.01))
# per CU: live workgroups over time and the turn-around: a workgroup's end -> the next workgroup start on that CU
wcu = cu_key[o][start][full]
o2 = np.lexsort((st, wcu))
c, s2, e2 = wcu[o2], st[o2], en[o2]
turn = []
for k in np.unique(c)[::8]:
    m = c == k
    ss, ee = np.sort(s2[m]), np.sort(e2[m])
    # the j-th end frees a slot; the first start after 5 (resident) workgroups that follows it
    res = 0
    d = np.zeros(ticks + 2, np.int64); np.add.at(d, ss, 1); np.add.at(d, ee, -1)
    l = np.cumsum(d)[lo:hi]
    turn.append(np.bincount(l, minlength=10)[:10])
turn = np.sum(turn, axis=0)
print('share of time a CU holds n live workgroups (steady part, every 8th CU): ' + ' '.join('%d:%.3f' % (n, turn[n] / turn.sum()) for n in range(10)))
