import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle, geosplatting_amd as gs
import geosplatting_amd.synthetic as syn
from geosplatting_amd import splitsum as ss
dev = torch.device('cuda:0')
def rel(a,b): return float(np.abs(a-b).max()/(np.abs(b).max()+1e-30))
g = torch.Generator().manual_seed(0)
cube = syn.make_cubemap(64, seed=2)
# mip fwd/bwd
x = cube.clone().to(dev).requires_grad_(True)
y = ss._CubeMapMip.apply(x)
print('mip fwd', rel(y.detach().cpu().numpy(), oracle.cubemap_mip_fwd(cube.numpy())))
v = torch.rand(6,32,32,3,generator=g)-0.5
y.backward(v.to(dev))
print('mip bwd', rel(x.grad.cpu().numpy(), oracle.cubemap_mip_bwd(v.numpy())))
# diffuse
c16 = torch.rand(6,16,16,3,generator=g)
x = c16.clone().to(dev).requires_grad_(True)
y = ss.diffuse_cubemap(x)
print('diffuse fwd', rel(y.detach().cpu().numpy(), oracle.diffuse_cubemap_fwd(c16.numpy())))
v = torch.rand(6,16,16,3,generator=g)-0.5
y.backward(v.to(dev))
print('diffuse bwd', rel(x.grad.cpu().numpy(), oracle.diffuse_cubemap_bwd(v.numpy())))
# specular per level
for R, rough in ((64,0.08),(32,0.5),(16,1.0)):
    c = torch.rand(6,R,R,3,generator=g)
    ct = oracle.ndf_cutoff(rough)
    b_ref = oracle.specular_bounds(R, ct)
    ct2, b = ss.specular_bounds(R, rough, 0.99, dev)
    print(R, rough, 'cutoff', ct, ct2, 'bounds equal', np.array_equal(b.cpu().numpy(), b_ref), 'nmismatch', (b.cpu().numpy()!=b_ref).sum())
    raw_ref = oracle.specular_cubemap_fwd(c.numpy(), b_ref, rough, ct)
    x = c.clone().to(dev).requires_grad_(True)
    y = ss.specular_cubemap(x, rough)
    ref = raw_ref[...,:3]/raw_ref[...,3:]
    print('  spec fwd', rel(y.detach().cpu().numpy(), ref))
    v = torch.rand(6,R,R,3,generator=g)-0.5
    y.backward(v.to(dev))
    gref = oracle.specular_cubemap_bwd(b_ref, (v.numpy()/raw_ref[...,3:]).astype(np.float32), rough, ct)
    print('  spec bwd', rel(x.grad.cpu().numpy(), gref))
