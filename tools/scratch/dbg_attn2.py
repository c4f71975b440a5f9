import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from oracle import oracle
from scalellm_amd import kernels
from scalellm_amd.decode import make_batch_inputs
DEV="cuda"
def run(H,HKV,D,q_lens,kv_lens,window=-1,alibi=False,B=16):
    _,_,p,nb = make_batch_inputs(q_lens, kv_lens, B, DEV, seed=3)
    g = torch.Generator(device=DEV).manual_seed(1)
    T=sum(q_lens)
    q = torch.randn(T,H,D,device=DEV,dtype=torch.bfloat16,generator=g)
    kc = torch.randn(nb*B,HKV,D,device=DEV,dtype=torch.bfloat16,generator=g)
    vc = torch.randn(nb*B,HKV,D,device=DEV,dtype=torch.bfloat16,generator=g)
    al = (torch.randn(H,device=DEV,generator=g)/max(kv_lens)).float() if alibi else None
    ref = oracle.paged_attn(q.float().cpu().numpy(), kc.float().cpu().numpy(), vc.float().cpu().numpy(), p.q_cu_seq_lens.cpu().numpy(), p.kv_cu_seq_lens.cpu().numpy(), p.block_tables.cpu().numpy(), p.cu_block_lens.cpu().numpy(), B, D**-0.5, 0.0, window, al.cpu().numpy() if alibi else None)
    for pf in (1,2):
        with kernels.tuning(SLM_ATTN_TILE_PF=pf):
            out = torch.full_like(q, float("nan"))
            kernels.paged_kv_varlen_mha(out,q,kc,vc,p.q_cu_seq_lens,p.kv_cu_seq_lens,p.block_tables,p.cu_block_lens,al,B,max(q_lens),max(kv_lens),D**-0.5,0.0,window)
            torch.cuda.synchronize()
        o = out.float().cpu().numpy()
        err = np.abs(np.nan_to_num(o, nan=1e9, posinf=1e9, neginf=-1e9)-ref).max(axis=(1,2))
        bad = np.where(err>2e-2)[0]
        print(dict(H=H,HKV=HKV,q=q_lens,kv=kv_lens,window=window,alibi=alibi,pf=pf), "max err", float(err.max()), "bad rows", bad[:12].tolist(), len(bad))
run(8,2,128,[100],[100])
run(8,2,128,[10],[10])
run(6,1,128,[10],[10])
run(6,1,128,[10],[10],alibi=True)
run(6,1,128,[10],[10],window=10)
run(8,2,128,[100],[300])
run(8,2,128,[70,33],[70,200],window=20)
run(32,8,128,[256],[1024])
