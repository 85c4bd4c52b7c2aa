import sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops, _lib as L
DEV = torch.device("cuda:0")
rng = np.random.default_rng(1)
pool = rng.integers(0, 12, size=(400000, 16)).astype(np.uint8)

def insert(seen, keys_np, base):
    k = torch.from_numpy(keys_np).to(DEV)
    n = k.shape[0]
    slots = torch.empty(n, dtype=torch.int32, device=DEV)
    L.call(DEV, L.lib().g2048_seen_insert, k.data_ptr(), L.u64(base), seen.table.data_ptr(), seen.capacity_log2,
           seen.count.data_ptr(), seen.overflow.data_ptr(), slots.data_ptr(), n, L.stream_ptr(DEV))
    torch.cuda.synchronize()
    return slots.cpu().numpy()

def table_keys(seen):
    t = seen.table.cpu().numpy().reshape(-1, 32)
    state = t[:, 24:28].copy().view(np.uint32).reshape(-1)
    keys = t[state == 2][:, :16]
    first = t[state == 2][:, 16:24].copy().view(np.uint64).reshape(-1)
    return state, keys, first

for trial, (cap0, n1, n2) in enumerate([(19, 196608, 196608), (12, 3000, 5000), (20, 300000, 300000)]):
    seen = ops.SeenStates(DEV, capacity_log2=cap0)
    idx1 = rng.integers(0, 250000, n1); idx2 = rng.integers(0, 400000, n2)
    seen.reserve(n1)
    s1 = insert(seen, pool[idx1], 0)
    u1 = len(np.unique(idx1))
    st, keys, first = table_keys(seen)
    print("trial", trial, "after insert1: count", int(seen.count.item()), "unique", u1, "overflow", int(seen.overflow.item()),
          "states", np.bincount(st, minlength=3)[:3], "distinct keys in table", len(np.unique(keys, axis=0)))
    old_log2 = seen.capacity_log2
    seen.overflow.zero_()
    seen.reserve(n2 + (1 << old_log2))        # force a rehash
    torch.cuda.synchronize()
    st, keys, first = table_keys(seen)
    print("   after rehash %d -> %d: overflow" % (old_log2, seen.capacity_log2), int(seen.overflow.item()), "states",
          np.bincount(st, minlength=3)[:3], "distinct keys", len(np.unique(keys, axis=0)))
    seen.overflow.zero_()
    s2 = insert(seen, pool[idx2], n1)
    u12 = len(np.unique(np.concatenate([idx1, idx2])))
    st, keys, first = table_keys(seen)
    print("   after insert2: count", int(seen.count.item()), "unique", u12, "overflow", int(seen.overflow.item()), "states",
          np.bincount(st, minlength=3)[:3], "distinct keys", len(np.unique(keys, axis=0)))
