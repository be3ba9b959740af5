"""Octree over the SDF zero set + lock-step sphere tracer.
utils/octree.py:19-57,124-199,217-265,377-438,459-471,493-592 and model/octree_tracing.py:31-60.

Node tables (all nodes, base grid first, then one block of 8 children per split parent, level by level):
  box_min[B,3], box_size[B,3] f32; child[B,8] i64 (-1 = none); is_split[B] bool;
  base_index[nx,ny,nz] i64; sdf_val[B]; sdf_nrm[B,3] (unit); centre[B,3]; hit[B] bool.
"""
import numpy as np
import torch

EPS_ADV = 1e-3


class OctreeTables:
    pass


def _octant_offsets():
    o = torch.arange(8)
    return torch.stack([(o // 4) % 2, (o // 2) % 2, o % 2], -1)          # [8,3] ints (octree.py:66-68)


def build(sdf_fn, grad_fn, box_min, box_max, cell=0.05, levels=4, thr=0.5, chunk=8192):
    """OctreeSDF.__init__ + Octree.build_base_grid/build (octree.py:377-409,124-181).
    sdf_fn: [M,3]->[M]; grad_fn: [M,3]->[M,3] (un-normalised input gradient of sdf_fn)."""
    T = OctreeTables()
    root_min = torch.tensor(box_min, dtype=torch.float32)
    root_size = torch.tensor([box_max[i] - box_min[i] for i in range(3)], dtype=torch.float32)
    ncell = (root_size / cell).ceil().long()
    root_size = (ncell * cell).float()                                    # octree.py:126-128
    T.root_min, T.root_size, T.ncell = root_min, root_size, ncell
    ax = [torch.arange(int(ncell[i])) for i in range(3)]
    anchor = torch.stack(torch.meshgrid(ax, indexing="ij"), -1).reshape(-1, 3)
    lo = (anchor / ncell) * root_size + root_min                          # into_box(inv=True) (octree.py:135-138)
    hi = ((anchor + 1.0) / ncell) * root_size + root_min
    mins, sizes = [lo], [hi - lo]
    child = [-torch.ones(lo.shape[0], 8, dtype=torch.long)]
    split = [torch.zeros(lo.shape[0], dtype=torch.bool)]
    T.base_index = torch.arange(lo.shape[0]).reshape(*[int(c) for c in ncell])
    start, total = 0, lo.shape[0]
    offs = _octant_offsets()
    for lvl in range(levels):
        bmin, bsz = mins[-1], sizes[-1]
        ctr = bmin + bsz * 0.5
        s = torch.cat([sdf_fn(ctr[j:j + chunk]) for j in range(0, ctr.shape[0], chunk)])
        div = s.abs() < bsz.norm(dim=-1) * thr                            # octree.py:381-385
        if not div.any():
            break
        k = div.nonzero()[:, 0]
        split[-1] = div
        first = total + 8 * torch.arange(k.shape[0])
        child[-1][k] = first[:, None] + torch.arange(8)[None, :]
        pm, ps = bmin[k][:, None, :], bsz[k][:, None, :]
        nmin = (pm + offs[None] * ps / 2).reshape(-1, 3)                  # divide() (octree.py:60-72)
        nsz = (ps / 2).expand(-1, 8, -1).reshape(-1, 3)
        mins.append(nmin); sizes.append(nsz)
        child.append(-torch.ones(nmin.shape[0], 8, dtype=torch.long))
        split.append(torch.zeros(nmin.shape[0], dtype=torch.bool))
        total += nmin.shape[0]
    T.box_min, T.box_size = torch.cat(mins), torch.cat(sizes)
    T.child, T.is_split = torch.cat(child), torch.cat(split)
    # combine_empty (octree.py:183-199) never changes is_split for these parameters (threshold > all sizes).
    T.centre = T.box_min + T.box_size * 0.5
    leaf = cell / 2 ** levels
    g = torch.cat([grad_fn(T.centre[j:j + chunk]) for j in range(0, total, chunk)])
    T.sdf_nrm = g / torch.clamp(g.norm(dim=-1, keepdim=True), min=1e-4)
    T.sdf_val = torch.cat([sdf_fn(T.centre[j:j + chunk]) for j in range(0, total, chunk)])
    T.min_step = float(torch.tensor(leaf, dtype=torch.float32)) + 1e-4
    T.hit = torch.relu(T.sdf_val) <= 1e-4
    return T


def locate(T, x):
    """Octree.query (octree.py:217-265): leaf index of each point, -1 when not strictly inside the root."""
    rel = (x - T.root_min) / T.root_size
    inside = ((rel < 1).all(-1)) & ((rel > 0).all(-1))
    out = -torch.ones(x.shape[0], dtype=torch.long)
    xi = x[inside]
    if xi.shape[0] == 0:
        return out
    res = torch.tensor(list(T.base_index.shape))
    ci = (((xi - T.root_min) / T.root_size) * res).floor().long()
    ptr = T.base_index[ci[:, 0], ci[:, 1], ci[:, 2]]
    k = T.is_split[ptr].nonzero()[:, 0]
    while k.numel() > 0:
        p = ptr[k]
        o = (((xi[k] - T.box_min[p]) / T.box_size[p]) * 2).long().clip(0, 1)   # truncation toward zero
        ptr[k] = T.child[p, 4 * o[:, 0] + 2 * o[:, 1] + o[:, 2]]
        k = T.is_split[ptr].nonzero()[:, 0]
    out[inside] = ptr
    return out


def slab(bmin, bsize, o, d):
    """intersect_box(forward_only=True) (octree.py:41-57) with IEEE 1/d."""
    inv = 1.0 / d
    ta, tb = (bmin - o) * inv, (bsize + bmin - o) * inv
    t1, t2 = torch.minimum(ta, tb), torch.maximum(ta, tb)
    near = t1.max(-1).values
    far = t2.min(-1).values
    return (near <= far) & (far >= 0), near.clamp(min=0.0), far


def _fine_march(T, pos, d, m, step):
    """fast_volume_render + first_nonzero (octree.py:459-471,588-592): marched distance until one step before
    the first sample whose cached cell SDF <= step; sample i sits at (i+2)*step; none -> (m+1)*step."""
    if pos.shape[0] == 0:
        return torch.zeros(0)
    t = torch.linspace(0, 1, m + 1) * m * step + step
    tm = t[1:]
    pts = pos[:, None, :] + d[:, None, :] * tm[:, None]
    ptr = locate(T, pts.reshape(-1, 3))
    s = T.sdf_val[ptr].reshape(-1, m)                                     # ptr == -1 reads the LAST node (octree.py:465-466)
    hit = torch.cat([s <= step, torch.ones(s.shape[0], 1, dtype=torch.bool)], -1)
    first = hit.float().argmax(-1)
    return t[first]


def cast(T, rays_o, rays_d, max_iter=-1, trace=None):
    """OctreeSDF.cast + multi_step_cast (octree.py:421-438,493-585) for a batch of R rays that share
    one lock-step schedule.  Returns (t[R], hit[R] bool).  trace (list) receives (n_active, m) per iteration."""
    R = rays_o.shape[0]
    o = rays_o + rays_d * 0.005 if max_iter > 0 else rays_o
    ok, near, _ = slab(T.root_min, T.root_size, o, rays_d)
    t = near + EPS_ADV
    t[~ok] = -1
    leaf = -torch.ones(R, dtype=torch.long)
    pos = torch.zeros_like(o)
    pos[ok] = o[ok] + t[ok, None] * rays_d[ok]
    if ok.any():
        leaf[ok] = locate(T, pos[ok])
    act = leaf >= 0
    it = 0
    while act.any():
        if max_iter > 0 and it > max_iter:
            break
        pk, dk, lk = pos[act], rays_d[act], leaf[act]
        _, _, far = slab(T.box_min[lk], T.box_size[lk], pk, dk)
        step = 0.001
        if max_iter > 0:
            step = 0.01 if R > 100000 else 0.005
        n_act = int(act.sum())
        m = int(np.clip(int(np.clip(R * 10, 1, 2000000) // n_act), 1, 100))
        if trace is not None:
            trace.append((n_act, m))
        small = far < m * step
        far = far.clone()
        far[small] = _fine_march(T, pk[small], dk[small], m, step)
        t[act] += far + EPS_ADV
        pos[act] = o[act] + t[act, None] * dk
        rel = (pos[act] - T.root_min) / T.root_size
        ins = torch.ones(R, dtype=torch.bool)
        ins[act] = ((rel < 1).all(-1)) & ((rel > 0).all(-1))
        act = act & ins
        leaf[~ins] = -1
        if act.any():
            leaf[act] = locate(T, pos[act])
            act = leaf >= 0
        if act.any():
            idx = act.nonzero()[:, 0]
            act[idx] = ~T.hit[leaf[idx]]
        it += 1
    hit = leaf >= 0                                                       # still-active rays count as hits
    if hit.any():
        lh = leaf[hit]
        n = T.sdf_nrm[lh]
        q = T.centre[lh] - n * T.sdf_val[lh].view(-1, 1)
        dist = ((q - pos[hit]) * n).sum(-1)
        speed = (rays_d[hit] * n).sum(-1)
        speed[speed == 0] = 1e-4
        t[hit] += torch.clamp(dist / speed, -T.min_step * 10, T.min_step * 10)
    return t, hit


def trace(T, cam_loc, ray_dirs, max_iter=-1, trace_log=None):
    """OctreeTracing.forward (octree_tracing.py:43-60): cam_loc [K,3], ray_dirs [K,P,3]
    -> x [K*P,3] (from the UN-offset origin), hit [K*P], t [K*P]."""
    o = cam_loc[:, None, :].expand(ray_dirs.shape).reshape(-1, 3)
    d = ray_dirs.reshape(-1, 3)
    t, hit = cast(T, o, d, max_iter, trace_log)
    return t[:, None] * d + o, hit, t


def octree_vis_logits(T, points, view_dirs):
    """OctreeVisModel.forward (model/octree_tracing.py:77-85): the secondary cast (max_iter = 32, set at :68) of ONE lock-step
    batch of rays `points + s*view_dirs`, returned as the float pair [is_hit, ~is_hit] that callers push through softmax
    (class 1 = visible, model/loss.py:174-177)."""
    _, hit = cast(T, points, view_dirs, 32)
    return torch.stack([hit, ~hit], -1).float()
