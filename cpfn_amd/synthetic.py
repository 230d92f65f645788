"""Seeded synthetic inputs of the shapes the reference's data loader hands to the
trainer (Utils/training_utils.py:122-132; Dataset normalisation
Utils/dataset_utils.py:26-27).  Used by bench.py, the tests and the golden
generator; CPU-generated so every rank / host reproduces the same bytes.
"""
import math

import torch


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def normalise_cloud(P):
    """Centre on the mean and scale so that max ||p|| = 1 (per cloud)."""
    P = P - P.mean(dim=1, keepdim=True)
    return P / P.norm(dim=2).max(dim=1, keepdim=True)[0].unsqueeze(2)


def uniform_cloud(B, N, seed=0):
    """P [B,N,3] uniform in the cube, normalised.  ~28 neighbours per r=0.2 ball."""
    g = _gen(seed)
    return normalise_cloud(torch.rand(B, N, 3, generator=g) * 2 - 1)


def _unit(v):
    return v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def primitive_cloud(B, N, n_prims=10, noise=0.005, seed=0):
    """Points sampled on `n_prims` random planes/spheres/cylinders/cones per cloud,
    plus Gaussian noise.  Returns dict(P, X_gt, I_gt, T_gt, axis) with
    T ids in the reference's config order sphere=0, plane=1, cylinder=2, cone=3
    (Configs/config_globalSPFN.yml:13-17) and axis [B,n_prims,3] the primitive's plane normal / cylinder or cone
    axis (an arbitrary unit vector for spheres).  Every label in [0, n_prims) is present.
    """
    g = _gen(seed)
    P = torch.empty(B, N, 3)
    X = torch.empty(B, N, 3)
    I = torch.empty(B, N, dtype=torch.long)
    T = torch.zeros(B, n_prims, dtype=torch.long)
    A = torch.zeros(B, n_prims, 3)
    for b in range(B):
        lab = torch.randint(0, n_prims, (N,), generator=g)
        lab[:n_prims] = torch.arange(n_prims)  # gap-free labels
        I[b] = lab
        for k in range(n_prims):
            m = (lab == k).nonzero().squeeze(1)
            n = m.numel()
            t = int(torch.randint(0, 4, (1,), generator=g))
            T[b, k] = t
            c = torch.rand(3, generator=g) * 1.2 - 0.6
            ax = _unit(torch.randn(3, generator=g))
            A[b, k] = ax
            e1 = _unit(torch.linalg.cross(ax, _unit(torch.randn(3, generator=g))))
            e2 = torch.linalg.cross(ax, e1)
            u = torch.rand(n, generator=g)
            v = torch.rand(n, generator=g)
            if t == 1:  # plane patch
                pts = c + (u[:, None] - 0.5) * 0.8 * e1 + (v[:, None] - 0.5) * 0.8 * e2
                nrm = ax.expand(n, 3)
            elif t == 0:  # sphere
                r = 0.15 + 0.25 * float(torch.rand(1, generator=g))
                d = _unit(torch.randn(n, 3, generator=g))
                pts = c + r * d
                nrm = d
            elif t == 2:  # cylinder
                r = 0.1 + 0.2 * float(torch.rand(1, generator=g))
                th = 2 * math.pi * u
                d = torch.cos(th)[:, None] * e1 + torch.sin(th)[:, None] * e2
                pts = c + r * d + (v[:, None] - 0.5) * 0.8 * ax
                nrm = d
            else:  # cone
                ha = 0.3 + 0.6 * float(torch.rand(1, generator=g))
                th = 2 * math.pi * u
                h = 0.1 + 0.5 * v
                d = torch.cos(th)[:, None] * e1 + torch.sin(th)[:, None] * e2
                pts = c + h[:, None] * ax + (h * math.tan(ha))[:, None] * d
                nrm = _unit(math.cos(ha) * d - math.sin(ha) * ax)
            P[b, m] = pts
            X[b, m] = nrm
        P[b] += noise * torch.randn(N, 3, generator=g)
    # normalise like the dataset does
    P = normalise_cloud(P)
    return {"P": P.contiguous(), "X_gt": X.contiguous(), "I_gt": I, "T_gt": T, "axis": A}


def training_batch(B, N=8192, n_max_instances=28, n_prims=10, n_inst_points=512,
                   kind="primitives", seed=0, consistent_axes=False):
    """One batch with every tensor spfn_train_val_epoch moves to the device
    (Utils/training_utils.py:122-132).  consistent_axes: the three GT axis tensors carry the primitives' own axes
    (what a data set provides; for runs that TRAIN on the batches) instead of random unit vectors (the default:
    bench.py, the fixtures and the tests only need the shapes)."""
    g = _gen(seed + 7919)
    if kind == "primitives":
        d = primitive_cloud(B, N, n_prims=n_prims, seed=seed)
        P, X_gt, I_gt = d["P"], d["X_gt"], d["I_gt"]
        T_gt = torch.zeros(B, n_max_instances, dtype=torch.long)
        T_gt[:, :n_prims] = d["T_gt"]
    else:
        P = uniform_cloud(B, N, seed=seed)
        X_gt = _unit(torch.randn(B, N, 3, generator=g))
        I_gt = torch.randint(0, n_prims, (B, N), generator=g)
        I_gt[:, :n_prims] = torch.arange(n_prims)
        T_gt = torch.randint(0, 4, (B, n_max_instances), generator=g)
    # points_per_instance: n_inst_points points of each GT instance (resampled with
    # replacement), zero rows for the unused instances
    ppi = torch.zeros(B, n_max_instances, n_inst_points, 3)
    for b in range(B):
        for k in range(n_prims):
            m = (I_gt[b] == k).nonzero().squeeze(1)
            sel = m[torch.randint(0, m.numel(), (n_inst_points,), generator=g)]
            ppi[b, k] = P[b, sel]
    axes = [_unit(torch.randn(B, n_max_instances, 3, generator=g)) for _ in range(3)]
    if consistent_axes and kind == "primitives":
        for a in axes:
            a[:, :n_prims] = d["axis"]
    return {
        "P": P.contiguous(), "X_gt": X_gt.contiguous(), "points_per_instance": ppi,
        "I_gt": I_gt, "T_gt": T_gt,
        "plane_n_gt": axes[0], "cylinder_axis_gt": axes[1], "cone_axis_gt": axes[2],
    }


def pointnet2_state_shapes(output_sizes=(3, 4, 28), use_glob_features=False, use_loc_features=False,
                           features_extractor=False):
    """state_dict keys -> shapes of the reference's PointNet2(dim_input=3, dim_pos=3, ...)
    (PointNet2/pn2_network.py:11-36), in construction order.  use_glob_features / use_loc_features widen sfp1's first
    layer by 1024 / 128 channels (:22-27); features_extractor drops bn1 and the heads (:31-36)."""
    shapes = {}
    extra = (1024 if use_glob_features else 0) + (128 if use_loc_features else 0)

    for name, cin, mlp in (("sa1", 3, (64, 64, 128)), ("sa2", 131, (128, 128, 256)),
                           ("sa3", 259, (256, 512, 1024))):
        # the reference registers all convs of a block before its batch-norms
        tmp = {}
        c = cin
        for j, cout in enumerate(mlp):
            tmp[j] = (c, cout)
            c = cout
        for j, (ci, co) in tmp.items():
            shapes["%s.conv_blocks.0.%d.weight" % (name, j)] = (co, ci, 1, 1)
            shapes["%s.conv_blocks.0.%d.bias" % (name, j)] = (co,)
        for j, (ci, co) in tmp.items():
            bn = "%s.bn_blocks.0.%d" % (name, j)
            shapes[bn + ".weight"] = (co,)
            shapes[bn + ".bias"] = (co,)
            shapes[bn + ".running_mean"] = (co,)
            shapes[bn + ".running_var"] = (co,)
            shapes[bn + ".num_batches_tracked"] = ()
    for name, cin, mlp in (("sfp1", 1280 + extra, (256, 256)), ("sfp2", 384, (256, 128)),
                           ("sfp3", 128, (128, 128, 128))):
        c = cin
        pairs = []
        for cout in mlp:
            pairs.append((c, cout))
            c = cout
        for j, (ci, co) in enumerate(pairs):
            shapes["%s.mlp_convs.%d.weight" % (name, j)] = (co, ci, 1)
            shapes["%s.mlp_convs.%d.bias" % (name, j)] = (co,)
        for j, (ci, co) in enumerate(pairs):
            bn = "%s.mlp_bns.%d" % (name, j)
            shapes[bn + ".weight"] = (co,)
            shapes[bn + ".bias"] = (co,)
            shapes[bn + ".running_mean"] = (co,)
            shapes[bn + ".running_var"] = (co,)
            shapes[bn + ".num_batches_tracked"] = ()
    shapes["fc1.weight"] = (128, 128, 1)
    shapes["fc1.bias"] = (128,)
    if features_extractor:
        return shapes
    for k in ("weight", "bias", "running_mean", "running_var"):
        shapes["bn1." + k] = (128,)
    shapes["bn1.num_batches_tracked"] = ()
    for j, o in enumerate(output_sizes):
        shapes["fc2.%d.weight" % j] = (o, 128, 1)
        shapes["fc2.%d.bias" % j] = (o,)
    return shapes


def synthetic_state_dict(shapes, seed=0):
    """Deterministic, RNG-order-independent weights keyed by parameter name, so the
    reference model, the oracle and the product model can all be loaded with the
    same tensors without shipping a 5.6 MB checkpoint.  Conv weights/biases are
    U(-1/sqrt(fan_in), +1/sqrt(fan_in)) (PyTorch's default scale); BatchNorm gammas
    are U(0.5, 1.5) with ~1 in 8 negated (exercises the sign-dependent pooling
    paths), betas U(-0.2, 0.2)."""
    import zlib
    out = {}
    for name, shape in shapes.items():
        g = _gen(zlib.crc32(name.encode()) + 1000003 * int(seed))
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            out[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            out[name] = torch.ones(shape)
        elif ".bn_blocks." in name or ".mlp_bns." in name or name.startswith("bn1."):
            if name.endswith("weight"):
                gam = torch.rand(shape, generator=g) + 0.5
                flip = torch.rand(shape, generator=g) < 0.125
                out[name] = torch.where(flip, -gam, gam)
            else:
                out[name] = torch.rand(shape, generator=g) * 0.4 - 0.2
        else:
            fan_in = shape[1] if len(shape) > 1 else None
            if fan_in is None:  # conv bias: fan_in of the matching weight
                fan_in = shapes[name[:-4] + "weight"][1]
            bound = 1.0 / math.sqrt(fan_in)
            out[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return out
