# kernel time of the consensus call for variant builds of the isoforms library (freddie_amd/variants/libfiso_<name>.so)
for v in "$@"; do for MM in ${MS:-150}; do
  FISO_LIB=$PWD/freddie_amd/variants/libfiso_$v.so timeout -k 5 120 python - <<P
import numpy as np, sys
sys.path.insert(0, '.')
from freddie_amd import isoforms
rng = np.random.default_rng(11)
n_iso, per, M = 4000, 500, $MM
R = n_iso * per
lab = rng.choice(np.frombuffer(b"0012", np.uint8), size=(R, M), p=[0.3, 0.3, 0.3, 0.1]).reshape(-1)
tail = rng.integers(0, 3, R).astype(np.uint8)
iro = np.arange(n_iso + 1, dtype=np.int64) * per
ctx = isoforms.Context(0)
off = np.arange(R, dtype=np.int64) * M
ms = []
for _ in range(6):
    ctx.kernel_ms = 0.0
    ctx.consensus(iro, np.full(n_iso, M), off, lab, tail)
    ms.append(ctx.kernel_ms)
pk = isoforms.pack_labels(lab)
mp = []
for _ in range(4):
    ctx.kernel_ms = 0.0
    ctx.consensus(iro, np.full(n_iso, M), off, pk, tail, packed=True)
    mp.append(ctx.kernel_ms)
print("$v M=$MM", "raw %.4f ms (%.3f of HBM)" % (min(ms[1:]), (n_iso * per * M + 8.0 * n_iso * M) / (min(ms[1:]) * 1e-3) / 8e12), "packed %.4f ms" % min(mp[1:]))
P
done; done
