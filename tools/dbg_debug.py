import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerind_amd as K
from tests import oracle as orc
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
name, k = sys.argv[1], int(sys.argv[2])
data = open(os.path.join(GOLD, name), "rb").read()
s = orc.kspec(k)
ctx = K.Context(0)
g = K.DeBruijnNodes(ctx, K.make_config(k))
ctx.profile(True)
g.build(data)
print({p["name"] for p in ctx.profile_get() if p["launches"]})
ok, oe = orc.dbg_parse(s, data)
om = orc.DbgMap(s); om.insert(ok, oe)
a = orc.sorted_rows(*[x if i == 0 else x.astype(np.uint64) for i, x in enumerate(g.to_vector())])
b = orc.sorted_rows(*[x if i == 0 else x.astype(np.uint64) for i, x in enumerate(om.export(canonical=True))])
print(a.shape, b.shape)
bad = np.nonzero((a != b).any(axis=1))[0]
print(len(bad), "rows differ")
for i in bad[:8]:
    print(hex(int(a[i, 0])), a[i, 1:].tolist(), "|", hex(int(b[i, 0])), b[i, 1:].tolist())
print(data[:400])
