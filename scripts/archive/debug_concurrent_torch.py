"""Do stock PyTorch elementwise kernels (built with packed-fp32 VALU instructions) keep their results beside conv_s2_kernel of
another stream?  (The library's own kernels did not until it was built without packed fp32: DESIGN.md.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd  # noqa
from oracle import sgg_oracle as O
from sgg_amd.lib import HipKernels
from sgg_amd.step import GanStep

K = HipKernels("cuda:0")
B, S, V = 8, 64, 50
images, labels, _ = O.synth_batch(B, S, V)
img, lab = images.cuda(), labels.cuda()
agg = GanStep(K, V, S, B, lam=10.0, g_state=O.init_params("G", V, S, perturb=0.05), d_state=O.init_params("D", V, S, perturb=0.05))
agg.critic_step(img, lab, O.synth_noise(B, 0).cuda(), O.synth_alpha(B, 0).reshape(B).cuda())
torch.cuda.synchronize()
T = agg.D.trunk
side = torch.cuda.Stream()


def beside(times=300):
    lay = T.layers[7]
    for _ in range(times):
        K.conv_fwd(T.layers[6]["a"], lay["w"], lay["w_fwd"], lay["b"], lay["y"], lay["s"], lay["ws_fwd"], T._am(0, 6), T._am(2, 7), lay["tstats"], lay["ws_layout"])


g = torch.Generator(device="cuda").manual_seed(3)
a, b, c = (torch.randn(1 << 22, device="cuda", generator=g) for _ in range(3))
m = torch.randn((512, 512), device="cuda", generator=g)
victims = {
    "add": lambda o: torch.add(a, b, out=o),
    "addcmul": lambda o: torch.addcmul(a, b, c, out=o),
    "mul_scalar_add": lambda o: torch.add(a, b, alpha=1.7, out=o),
    "lerp": lambda o: torch.lerp(a, b, 0.3, out=o),
    "fma chain": lambda o: o.copy_(a).mul_(b).add_(c).mul_(1.0001).add_(b),
    "sum(dim)": lambda o: torch.sum(a.view(4096, 1024), dim=1, out=o[:4096]),
    "adam-like": lambda o: o.copy_(a).abs_().mul_(0.9).addcmul_(b, b, value=0.1).sqrt_().add_(1e-8),
}
for name, fn in victims.items():
    ref = torch.zeros_like(a)
    fn(ref)
    torch.cuda.synchronize()
    bad = nbad = 0
    overl = True
    for rep in range(15):
        outs = [torch.zeros_like(a) for _ in range(8)]
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            beside()
        for o in outs:
            fn(o)
        overl &= not side.query()
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                nbad += int((o != ref).sum())
    print("torch victim %-16s beside conv_s2: %d of 120 outputs differ (%d elements); overlapped %s" % (name, bad, nbad, overl), flush=True)
