"""`model.refinenet.RefineNet` (reference model/refinenet.py:5-37): 75->160->256->256->128->45
MLP with BatchNorm1d + ReLU.  The module owns the 30 checkpoint keys (`block.layerN.{0,1}.*`,
`block.layer5.*`); `folded(device)` returns the BN-folded, transposed weights the HIP
`smap_refine` kernel consumes; `forward` runs that kernel on [N,75] inputs."""
import torch
import torch.nn as nn

DIMS = (75, 160, 256, 256, 128, 45)


class RefineNet_base(nn.Module):
    def __init__(self, in_dim=75, out_dim=45, flatten_size=1):
        super().__init__()
        if (in_dim, out_dim, flatten_size) != (75, 45, 1):
            raise NotImplementedError("the HIP RefineNet kernel is specialised to 75->45")
        for l in range(4):
            setattr(self, f"layer{l + 1}", nn.Sequential(nn.Linear(DIMS[l], DIMS[l + 1]),
                                                         nn.BatchNorm1d(DIMS[l + 1]), nn.ReLU()))
        self.layer5 = nn.Linear(DIMS[4], DIMS[5])
        self.out_dim = out_dim


class RefineNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.block = RefineNet_base()

    @torch.no_grad()
    def folded(self, device):
        """(wt[5], bs[5]): BN folded (eval), weights transposed to [in][out], fp32 on `device`."""
        wt, bs = [], []
        for l in range(4):
            lin, bn = getattr(self.block, f"layer{l + 1}")[0], getattr(self.block, f"layer{l + 1}")[1]
            s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            w = lin.weight.double() * s[:, None]
            b = (lin.bias.double() - bn.running_mean.double()) * s + bn.bias.double()
            wt.append(w.t().contiguous().float().to(device))
            bs.append(b.float().to(device))
        wt.append(self.block.layer5.weight.detach().t().contiguous().float().to(device))
        bs.append(self.block.layer5.bias.detach().float().to(device))
        return wt, bs

    def forward(self, input_x):
        """input_x [N,75] (2D pose + root-relative 3D pose) -> [N,45] on input_x's device."""
        from .. import dapalib
        if self.training:
            raise RuntimeError("smap_amd.RefineNet runs eval-mode inference only")
        if not input_x.is_cuda:
            raise RuntimeError("smap_amd.RefineNet.forward needs a ROCm GPU tensor; there is no CPU fallback")
        return dapalib.refine_mlp(input_x.float().contiguous(), *self.folded(input_x.device))
