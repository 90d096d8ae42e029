"""Random attention-layer configurations (kind, channels, sequence length, the free extent, axis, stride, batch, BatchNorm groups,
mode) through the product path on the CPU lane emulator against float64 autograd through the oracle -- a bug hunt over shapes no
fixture covers.   python scripts/emu_layer_hunt.py <seed> <count>      (no GPU; ~2 s per configuration)"""
import sys, random, traceback; import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'medical-transformer_amd'), ROOT]
import ctypes as C, numpy as np, torch
import test_lane_emu as T, test_axial_layer_gpu as TG
from emu_device import emulated_device
from medt_amd import _lib as L
from oracle import medt_oracle as O
lib = C.CDLL(T.build_emulator())
for name,(res,args) in L.SIGNATURES.items():
    fn=getattr(lib,name); fn.restype,fn.argtypes=res,args
rng = random.Random(int(sys.argv[1]))
n_ok=n_bad=0
for it in range(int(sys.argv[2])):
    kind = rng.choice(["dynamic","plain","wopos","gatedsig","gateddata","dynamic"])
    Cc = rng.choice([16,32,64,128])
    Lq = rng.choice([2,4,6,8,12,16,20,32,48,64])
    other = rng.choice([1,2,3,4,5,6,7,8,10])
    width = rng.random()<0.5
    stride = rng.choice([1,1,2])
    if stride==2 and (Lq%2 or other%2): continue
    N = rng.choice([1,2,3,4,5])
    groups = rng.choice([g for g in (1,2,N) if N%g==0])
    training = rng.random()<0.7
    if training and (N//groups)*other*Lq < 2: continue
    cfg=(kind,Cc,Lq,other,width,stride,N,groups,training)
    try:
        layer = TG.make_layer(kind, Cc, Lq, width, stride, "cpu")
        st = O.randomize_state({k: v.clone() for k, v in layer.state_dict().items()}, 77)
        g = torch.Generator().manual_seed(it)
        Hh, Ww = (other, Lq) if width else (Lq, other)
        x = torch.randn((N, Cc, Hh, Ww), generator=g).double()
        dout = torch.randn((N, Cc, Hh//stride, Ww//stride), generator=g).double()
        with emulated_device(lib):
            got, want = TG.run_case(layer, st, x, dout, kind, width, stride, "cpu", training, groups)
        TG.compare(got, want, tol=1e-3)
        n_ok+=1
    except Exception as e:
        n_bad+=1
        print("FAIL", cfg, type(e).__name__, str(e)[:300].replace("\n"," "))
print("ok", n_ok, "bad", n_bad)
