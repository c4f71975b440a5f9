import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from tests import helpers
from tests import test_attention_gpu as t
GOLD = helpers.load_attn_cases()
for name in ("mqa_alibi_window", "mixed_softcap", "spec_verify", "mha_d96_window0"):
    c = GOLD[name]
    dtype = torch.bfloat16 if c["is_bf16"] else torch.float16
    case = dict(q=c["q_f32"], key_cache=c["k_f32"], value_cache=c["v_f32"], q_cu_lens=c["q_cu_lens"], kv_cu_lens=c["kv_cu_lens"],
                block_table=c["block_table"], block_cu_lens=c["block_cu_lens"], block_size=c["block_size"],
                max_q_len=int(np.diff(c["q_cu_lens"]).max()), max_kv_len=int(np.diff(c["kv_cu_lens"]).max()))
    for pf in (1, 2):
        from scalellm_amd import kernels
        with kernels.tuning(SLM_ATTN_TILE_PF=pf):
            out, _ = t._run_hip(case, dtype, c["sm_scale"], c["softcap"], c["window"], c["alibi"])
        nanrows = np.unique(np.argwhere(np.isnan(out))[:, 0]).tolist()
        err = np.abs(np.nan_to_num(out) - c["out"]).max(axis=(1, 2))
        print(name, "pf", pf, "nan rows:", nanrows, "q_cu", c["q_cu_lens"].tolist(), "kv", np.diff(c["kv_cu_lens"]).tolist(), "window", c["window"], "max err per row", np.round(err, 3).tolist())
