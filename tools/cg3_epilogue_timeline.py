import ctypes as C, sys
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from naturaldiffusion_amd._lib import lib, check, ptr, stream_ptr
for res, B, cin, N, c1, resid in ((32,512,128,128,0,0),(32,512,128,128,0,1),(16,512,256,256,0,0),(16,512,256,256,0,1)):
    dev="cuda"; M=B*res*res
    x=torch.randn(B,res,res,cin,device=dev).bfloat16(); sc=torch.rand(B,cin,device=dev)+0.5; sh=torch.randn(B,cin,device=dev)*0.3
    w=(torch.randn(N,9*cin+c1,device=dev)/(9*cin)**0.5).bfloat16(); bias=torch.randn(N,device=dev); out=torch.empty(M,N,dtype=torch.bfloat16,device=dev)
    part=torch.zeros(M//128,N//4,2,device=dev); wf=torch.zeros_like(w)
    ts=torch.zeros(128,dtype=torch.int64,device=dev)
    f=lib.natinf_debug_cg3_timeline; f.restype=C.c_int; f.argtypes=[C.c_void_p]; check(f(ptr(ts)),"tl")
    check(lib.natinf_set_conv_gn_w128(7),"k")
    for s_ in range(3): check(lib.natinf_set_conv_gn_w128_min_k(s_,0),"m")
    args=(res,B,N,cin,c1,ptr(x),ptr(sc),ptr(sh),ptr(w),ptr(wf),None,ptr(bias),None,0.7071,ptr(out),ptr(part))
    for _ in range(3): check(lib.natinf_debug_conv_gn(*args,1,stream_ptr()),"run"); torch.cuda.synchronize()
    t=ts[:64].tolist()
    print(f"res {res} cin {cin} N {N}: epilogue (cg3 stamps) {t[43]-t[42]}: entry->packed start {t[2]-t[42]}, register phase + slab + barrier {t[3]-t[2]}, copy-out {t[4]-t[3]}, partials tail {t[43]-t[4]}")
