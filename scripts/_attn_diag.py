import os, sys
os.environ["REVO_EXPERIMENTS"]="1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import _lib
lib=_lib.load(); dev=torch.device("cuda",0)
for (B,S,H) in [(64,577,16)]:
    hd=64; W=H*hd
    g=torch.Generator(device="cpu").manual_seed(S*31+H)
    qkv=torch.randn(B*S,3*W,generator=g).to(dev).bfloat16()
    def run(flag):
        lib.revo_op_set_variant(flag)
        out=torch.full((B*S,W),float("nan"),device=dev,dtype=torch.bfloat16)
        _lib.check(lib.revo_op_attention(_lib.ptr(qkv),3*W,_lib.ptr(out),W,B,S,H,hd,_lib.current_stream()))
        torch.cuda.synchronize()
        return out
    old=run(0)
    bad=[]
    outs=[run(1<<20) for _ in range(6)]
    for k,o in enumerate(outs):
        d=(o.float()-old.float()).abs().view(B,S,H,hd).amax(-1)      # per (image, token, head)
        idx=(d>0.02).nonzero()
        bad.append([tuple(x) for x in idx.tolist()][:6])
    same=[bool(torch.equal(outs[0],o)) for o in outs[1:]]
    print(os.environ.get("REVO_ATTN16_DBG"), (B,S,H), "repeat-identical:", same, "n differing elems vs run0:", [int((o!=outs[0]).sum()) for o in outs[1:]], flush=True)
lib.revo_op_set_variant(0)
