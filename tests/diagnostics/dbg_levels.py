import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from oracle import nets, seeded
from learned_hologram_gan_amd.neural_network_components import UNet
from learned_hologram_gan_amd import hip_ops as ops
from learned_hologram_gan_amd.hip_ops import OutSlot
def rel(a,b): return ((a-b).abs().max()/b.abs().max()).item()
rows,batch=64,4
sd32 = {k[len("part1.part1."):]: v for k, v in seeded.generator_state_dict().items() if k.startswith("part1.part1.")}
rgbd, _, _ = seeded.smooth_batch(batch, rows, rows, seed=21)
proj = torch.randn((batch, 6, rows, rows), generator=torch.Generator().manual_seed(8))
# ---- oracle with retained intermediates (fp64)
def oracle(dtype):
    sd = nets.as_parameters({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd32.items()})
    keep = {}
    blk = lambda name, t: nets.residual_block(sd, name, t, True)
    up = lambda name, t: F.conv_transpose2d(t, sd[name + ".weight"], sd[name + ".bias"], stride=2)
    pool = lambda t: F.max_pool2d(t, 2, 2)
    def k(name, t): t.retain_grad(); keep[name] = t; return t
    x = rgbd.to(dtype)
    e1 = k("e1", blk("encoder1.0.0", x)); p1 = k("p1", pool(e1))
    e2 = k("e2", blk("encoder2.1.0", p1)); p2 = k("p2", pool(e2))
    e3 = k("e3", blk("encoder3.1.0", p2)); p3 = k("p3", pool(e3))
    e4 = k("e4", blk("encoder4.1.0", p3)); p4 = k("p4", pool(e4))
    b = k("b", blk("bottleneck.1.0", p4)); u1 = k("u1", up("bottleneck.2", b))
    c1 = k("c1", torch.cat((e4, u1), 1)); d1 = k("d1", blk("decoder1.0.0", c1)); u2 = k("u2", up("decoder1.1", d1))
    c2 = k("c2", torch.cat((e3, u2), 1)); d2 = k("d2", blk("decoder2.0.0", c2)); u3 = k("u3", up("decoder2.1", d2))
    c3 = k("c3", torch.cat((e2, u3), 1)); d3 = k("d3", blk("decoder3.0.0", c3)); u4 = k("u4", up("decoder3.1", d3))
    c4 = k("c4", torch.cat((e1, u4), 1)); d4 = k("d4", blk("decoder4.0", c4))
    y = torch.sigmoid(F.conv2d(d4, sd["final_layer.0.weight"], sd["final_layer.0.bias"]))
    (y * proj.to(dtype)).sum().backward()
    return {n: t.grad.double() for n, t in keep.items()}, {n: t.detach().double() for n, t in keep.items()}
g64, v64 = oracle(torch.float64); g32, v32 = oracle(torch.float32)
# ---- product, mirrored forward
net = UNet(6, 4); net.load_state_dict(sd32); net.to("cuda").train()
X = rgbd.to("cuda"); N, _, H, W = X.shape; dev = X.device
keep = {}
def k(name, t): t.retain_grad(); keep[name] = t; return t
x = ops.ToNHWC.apply(X, 32)
new = lambda h, w, c: torch.empty((N, h, w, c), dtype=torch.float32, device=dev)
buf4, buf3, buf2, buf1 = new(H, W, 128), new(H // 2, W // 2, 256), new(H // 4, W // 4, 512), new(H // 8, W // 8, 1024)
pool = ops.MaxPool2x2Fn.apply; B = net._block
e1 = k("e1", B(net.encoder1[0]).forward_nhwc(x, OutSlot(buf4[..., :64]))); p1 = k("p1", pool(e1))
e2 = k("e2", B(net.encoder2[1]).forward_nhwc(p1, OutSlot(buf3[..., :128]))); p2 = k("p2", pool(e2))
e3 = k("e3", B(net.encoder3[1]).forward_nhwc(p2, OutSlot(buf2[..., :256]))); p3 = k("p3", pool(e3))
e4 = k("e4", B(net.encoder4[1]).forward_nhwc(p3, OutSlot(buf1[..., :512]))); p4 = k("p4", pool(e4))
b = k("b", B(net.bottleneck[1]).forward_nhwc(p4)); u1 = k("u1", net._up(net.bottleneck[2], b, OutSlot(buf1[..., 512:])))
c1 = k("c1", net._cat(e4, u1, buf1)); d1 = k("d1", B(net.decoder1[0]).forward_nhwc(c1)); u2 = k("u2", net._up(net.decoder1[1], d1, OutSlot(buf2[..., 256:])))
c2 = k("c2", net._cat(e3, u2, buf2)); d2 = k("d2", B(net.decoder2[0]).forward_nhwc(c2)); u3 = k("u3", net._up(net.decoder2[1], d2, OutSlot(buf3[..., 128:])))
c3 = k("c3", net._cat(e2, u3, buf3)); d3 = k("d3", B(net.decoder3[0]).forward_nhwc(c3)); u4 = k("u4", net._up(net.decoder3[1], d3, OutSlot(buf4[..., 64:])))
c4 = k("c4", net._cat(e1, u4, buf4)); d4 = k("d4", B(net.decoder4).forward_nhwc(c4))
head = net.final_layer[0]; y = ops.SigmoidHeadFn.apply(d4, head.weight, head.bias)
(y * proj.to("cuda")).sum().backward()
for n in ["d4","c4","u4","d3","c3","u3","d2","c2","u2","d1","c1","u1","b","p4","e4","p3","e3","p2","e2","p1","e1"]:
    gg = keep[n].grad.detach().cpu().permute(0,3,1,2).double(); vv = keep[n].detach().cpu().permute(0,3,1,2).double()
    print("%-3s value gpu %.1e | grad cpu32 %.2e gpu %.2e" % (n, rel(vv, v64[n]), rel(g32[n], g64[n]), rel(gg, g64[n])))
print("---- e3 decomposition")
ge3 = keep["e3"].grad; gp3 = keep["p3"].grad; gc2 = keep["c2"].grad
e3v = keep["e3"].detach()
# reference pool backward with torch on the GPU
e3n = e3v.permute(0,3,1,2).contiguous().requires_grad_(True)
pp = F.max_pool2d(e3n, 2, 2); pp.backward(gp3.permute(0,3,1,2).contiguous())
pool_ref = e3n.grad.permute(0,2,3,1)
mine = ops.MaxPool2x2Fn.apply(e3v.clone().requires_grad_(True))
xx = e3v.clone().requires_grad_(True); yy = ops.MaxPool2x2Fn.apply(xx); yy.backward(gp3.contiguous()); pool_mine_dense = xx.grad
print("pool bwd (dense x) vs torch:", rel(pool_mine_dense.double().cpu(), pool_ref.double().cpu()))
xs = buf2[..., :256].detach().requires_grad_(True); ys = ops.MaxPool2x2Fn.apply(xs); ys.backward(gp3.contiguous()); pool_mine_strided = xs.grad
print("pool bwd (strided x) vs torch:", rel(pool_mine_strided.double().cpu(), pool_ref.double().cpu()))
tot = pool_ref + gc2[..., :256]
print("e3.grad vs (pool_ref + c2.grad[:256]):", rel(ge3.double().cpu(), tot.double().cpu()))
print("e3.grad vs pool_ref only:", rel(ge3.double().cpu(), pool_ref.double().cpu()), " vs slice only:", rel(ge3.double().cpu(), gc2[..., :256].double().cpu()))
print("strides: e3.grad", ge3.stride(), "c2.grad", gc2.stride(), "p3.grad", gp3.stride())
print("---- c2 halves")
gc = gc2.detach().cpu().permute(0,3,1,2).double()
for name, sl in (("first(e3)", slice(0,256)), ("second(u2)", slice(256,512))):
    print(name, "gpu vs f64 %.3e | cpu32 vs f64 %.3e | max|g64| %.3e" % (rel(gc[:, sl], g64["c2"][:, sl]), rel(g32["c2"][:, sl], g64["c2"][:, sl]), g64["c2"][:, sl].abs().max().item()))
for name in ("c1", "c3", "c4"):
    g = keep[name].grad.detach().cpu().permute(0,3,1,2).double(); C = g.shape[1] // 2
    print(name, "first gpu %.3e cpu %.3e | second gpu %.3e cpu %.3e" % (rel(g[:, :C], g64[name][:, :C]), rel(g32[name][:, :C], g64[name][:, :C]), rel(g[:, C:], g64[name][:, C:]), rel(g32[name][:, C:], g64[name][:, C:])))
print("---- argmax flips")
for name in ("e1","e2","e3","e4"):
    vg = keep[name].detach().cpu().permute(0,3,1,2).double(); vo = v64[name]
    _, ig = F.max_pool2d(vg, 2, 2, return_indices=True); _, io = F.max_pool2d(vo, 2, 2, return_indices=True)
    _, i32 = F.max_pool2d(v32[name], 2, 2, return_indices=True)
    flips = (ig != io); flips32 = (i32 != io)
    # how many of the flipped windows are all-zero (harmless) vs positive
    mx = F.max_pool2d(vo, 2, 2)
    print(name, "windows", ig.numel(), "gpu flips", int(flips.sum()), "with max>0:", int((flips & (mx > 0)).sum()), "| cpu32 flips", int(flips32.sum()), "with max>0:", int((flips32 & (mx>0)).sum()))
