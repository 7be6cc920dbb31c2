"""Numerical side of the Winograd question (VERDICT r2 task 3c): how much forward noise would F(2,3) (1-D, 1.5 x fewer
multiplies) and F(2x2,3x3) (2.25 x fewer) add over a direct fp32 convolution on VGG-like data? All three evaluated in
fp32 (transforms and sums in fp32, as a kernel with fp32 accumulators would) against an fp64 direct convolution. The
forward noise sets the max-pool argmax-flip fraction of the texture gradient. CPU only."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def wino2d(x, w):   # x [C,H,W] fp32 (H, W even), w [O,C,3,3] -> [O,H,W], all arithmetic fp32
    C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(1, 4, 2).unfold(2, 4, 2)                                  # [C, H/2, W/2, 4, 4]
    bt, g, at = BT.float(), G.float(), AT.float()
    V = torch.einsum("ij,cyxjk,lk->cyxil", bt, d, bt)                        # B^T d B
    U = torch.einsum("ij,ocjk,lk->ocil", g, w, g)                            # G g G^T (fp64 on the host in a product)
    M = torch.einsum("ocil,cyxil->oyxil", U, V)
    Y = torch.einsum("ij,oyxjk,lk->oyxil", at, M, at)                        # [O, H/2, W/2, 2, 2]
    return Y.permute(0, 1, 3, 2, 4).reshape(w.shape[0], H, W)


def wino1d(x, w):   # F(2,3) along x, direct along y
    C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2)                                                   # [C, H+2, W/2, 4]
    bt, g, at = BT.float(), G.float(), AT.float()
    V = torch.einsum("ij,cyxj->cyxi", bt, d)
    U = torch.einsum("ij,ockj->ocki", g, w)                                  # [O,C,ky,4]
    M = sum(torch.einsum("oci,cyxi->oyxi", U[:, :, ky], V[:, ky:ky + H]) for ky in range(3))
    Y = torch.einsum("ij,oyxj->oyxi", at, M)                                 # [O,H,W/2,2]
    return Y.reshape(w.shape[0], H, W)


print("layer shape        direct fp32   F(2,3) 1-D   F(2x2,3x3)   (rms error / rms of the fp64 result; ratio to direct)")
for C, O, H, W in [(64, 64, 64, 64), (128, 128, 48, 48), (256, 256, 32, 32), (512, 512, 16, 16)]:
    x = F.relu(torch.randn(C, H, W) * 3)
    w = torch.randn(O, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    ref = F.conv2d(x.double()[None], w.double(), padding=1)[0]
    rms = float((ref ** 2).mean().sqrt())
    e = {}
    e["direct"] = float(((F.conv2d(x[None], w, padding=1)[0].double() - ref) ** 2).mean().sqrt()) / rms
    e["w1"] = float(((wino1d(x, w).double() - ref) ** 2).mean().sqrt()) / rms
    e["w2"] = float(((wino2d(x, w).double() - ref) ** 2).mean().sqrt()) / rms
    print(f"{C:3d}->{O:3d} {H:3d}x{W:<3d}   {e['direct']:.2e}     {e['w1']:.2e} ({e['w1'] / e['direct']:.1f}x)   "
          f"{e['w2']:.2e} ({e['w2'] / e['direct']:.1f}x)")
