import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests import helpers
import tests.test_attention_gpu as T
cfgs = T._grid_sample(72, seed=11)
for i, cfg in enumerate(cfgs):
    if not (cfg["batch"] == 4 and cfg["block"] == 8 and cfg["max_q"] == 125 and cfg["max_kv"] == 127 and cfg["kv_heads"] == 1 and cfg["softcap"] == 50.0 and cfg["alibi"]):
        continue
    case = helpers.make_paged_case(1000 + i, cfg["batch"], cfg["max_q"], cfg["max_kv"], 6, cfg["kv_heads"], cfg["head_dim"], cfg["block"], unique_blocks=True)
    alibi = (np.random.default_rng(i).standard_normal(6) / cfg["max_kv"]).astype(np.float32)
    sm = cfg["head_dim"] ** -0.5
    out, rounded = T._run_hip(case, cfg["dtype"], sm, cfg["softcap"], cfg["window"], alibi)
    ref = T._oracle(case, rounded, sm, cfg["softcap"], cfg["window"], alibi)
    qcu, kcu = case["q_cu_lens"], case["kv_cu_lens"]
    print("cfg", i, cfg, "q_lens", np.diff(qcu), "kv_lens", np.diff(kcu))
    for b in range(len(qcu) - 1):
        o, r = out[qcu[b]:qcu[b+1]], ref[qcu[b]:qcu[b+1]]
        err = np.abs(o - r).max(axis=(1, 2)) if o.size else np.zeros(0)
        bad = np.nonzero(err > 0.02)[0]
        print(" seq", b, "rows", o.shape[0], "bad tokens", len(bad), (bad[:10], bad[-5:]) if len(bad) else "")
        if len(bad):
            t = bad[0]
            print("   token", t, "per-head err", np.abs(o[t]-r[t]).max(axis=1))
    np.set_printoptions(linewidth=250, precision=3, suppress=True)
    for b in (0, 2):
        o, r = out[qcu[b]:qcu[b+1]], ref[qcu[b]:qcu[b+1]]
        e = np.abs(o - r).max(axis=2).reshape(-1)   # row = token*6 + head
        print("seq", b, "row errors (first 160 rows):")
        print((e[:160] > 0.02).astype(int).reshape(-1, 32))
        print("tail rows bad count per 32:", (e > 0.02).astype(int)[: (len(e)//32)*32].reshape(-1, 32).sum(axis=1))
    break
