import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/send-slam_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from send_slam_amd import binding
from oracle import orb_oracle as O
n_parts, nq, n_db = 8, 2000, 160000
rng = np.random.default_rng(n_db)
q = rng.integers(0, 256, size=(nq, 32), dtype=np.uint8)
db = rng.integers(0, 256, size=(n_db, 32), dtype=np.uint8)
per = (n_db + n_parts - 1) // n_parts
db[3] = q[0]; db[per + 5] = q[0]; db[per + 9] = q[1]; db[7] = q[2]; db[7, 0] ^= 1; db[per + 1] = q[2]; db[per + 1, 5] ^= 3
dev = torch.device("cuda:0")
tq, tdb = torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev)
parts = torch.empty((n_parts, nq), dtype=torch.int64, device=dev)
idx = torch.empty(nq, dtype=torch.int32, device=dev); d1 = torch.empty(nq, dtype=torch.int16, device=dev); d2 = torch.empty(nq, dtype=torch.int16, device=dev)
want = O.match(q, db, th=-1, ratio_num=1)
rawwant = [O.match(q, db[r*per:(r+1)*per], th=-1) for r in range(n_parts)]
with binding.OrbContext(0) as ctx:
    for rep in range(30):
        for r in range(n_parts):
            b, e = min(r * per, n_db), min((r + 1) * per, n_db)
            ctx.match_partial_device(tq.data_ptr(), nq, tdb[b:e].data_ptr(), e - b, b, parts[r].data_ptr())
        ctx.synchronize()
        rec = parts.cpu().numpy().view(np.dtype([("d1", "<u2"), ("d2", "<u2"), ("row", "<i4")]))
        for r in range(n_parts):
            w = rawwant[r]
            bad = np.nonzero((rec["d1"][r] != w[1]) | (rec["d2"][r] != w[2]) | (rec["row"][r] != w[0] + r * per))[0]
            if len(bad):
                i = bad[0]
                print("rep", rep, "part", r, "n_bad", len(bad), "query", i, "got", rec[r][i], "want", w[0][i] + r*per, w[1][i], w[2][i])
print("done")
