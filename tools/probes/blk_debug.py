import ctypes, torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from airpose_amd import _native as Nn
L = Nn.lib(); dev = torch.device("cuda", 0)
bf = torch.bfloat16; N, H = 2, 14
g = torch.Generator().manual_seed(5)
x = torch.randn(N, H, H, 1024, generator=g).to(bf).to(dev)
w1 = (torch.randn(256, 1024, generator=g) * (2.0 / 1024) ** 0.5).to(bf).to(dev)
w2 = (torch.randn(256, 2304, generator=g) * (2.0 / 2304) ** 0.5).to(bf).to(dev)
w3 = (torch.randn(1024, 256, generator=g) * (2.0 / 256) ** 0.5).to(bf).to(dev)
sc = [(torch.rand(c, generator=g) + 0.5).to(dev) for c in (256, 256, 1024)]
sh = [(torch.randn(c, generator=g) * 0.1).to(dev) for c in (256, 256, 1024)]
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = Nn.stream_ptr(dev); B = Nn.PRECISIONS["bf16"]
ws = torch.empty(L.ap_block_img_stream_bytes(), dtype=torch.uint8, device=dev)
Nn.check(L.ap_block_img_pack(B, p(w1), p(w2), p(w3), p(ws), st), "pack")
y = torch.full((N, H, H, 1024), float("nan"), dtype=bf, device=dev)
Nn.check(L.ap_block_img_nhwc(B, p(x), p(ws), p(sc[0]), p(sh[0]), p(sc[1]), p(sh[1]), p(sc[2]), p(sh[2]), p(y), N, st), "blk")
t1 = torch.empty(N, H, H, 256, dtype=bf, device=dev); t2 = torch.empty_like(t1); y2 = torch.empty_like(y)
L.ap_set_conv_config(11)
Nn.check(L.ap_conv2d_nhwc(B, p(x), p(w1), p(sc[0]), p(sh[0]), None, p(t1), N, H, H, 1024, 256, 1, 1, 0, 1, st), "c1")
Nn.check(L.ap_conv2d_nhwc(B, p(t1), p(w2), p(sc[1]), p(sh[1]), None, p(t2), N, H, H, 256, 256, 3, 1, 1, 1, st), "c2")
Nn.check(L.ap_conv2d_nhwc(B, p(t2), p(w3), p(sc[2]), p(sh[2]), p(x), p(y2), N, H, H, 256, 1024, 1, 1, 0, 1, st), "c3")
L.ap_set_conv_config(-1)
torch.cuda.synchronize()
bad = (y.view(torch.int16) != y2.view(torch.int16)).cpu()
print("bad frac", bad.float().mean().item())
print("by image", bad.float().mean((1, 2, 3)))
print("by row", bad.float().mean((0, 2, 3)))
print("by col", bad.float().mean((0, 1, 3)))
bc = bad.float().mean((0, 1, 2))
print("by channel chunk of 32:", bc.view(32, 32).mean(1))
print("by channel mod 32:", bc.view(32, 32).mean(0))
d = (y.float() - y2.float()).abs().cpu()
print("max abs diff", d.max().item(), "ref max", y2.float().abs().max().item())
