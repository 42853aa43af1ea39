"""The oracle (oracle/mtlora_oracle.py) against the golden vectors captured from the real
reference by tests/golden/make_golden.py.  CPU only.  Tolerances: fp64 1e-10."""
import pytest
import torch

from oracle import mtlora_oracle as O

TOL = dict(rtol=1e-9, atol=1e-10)


def close(a, b, **kw):
    kw = {**TOL, **kw}
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, **kw), (a - b).abs().max().item()


def check_sum(t, cs, rtol=1e-8):
    f = t.detach().double().flatten()
    assert tuple(t.shape) == tuple(cs["shape"])
    assert abs(f.sum().item() - cs["sum"]) <= rtol * max(1.0, cs["abssum"])
    assert abs(f.abs().sum().item() - cs["abssum"]) <= rtol * max(1.0, cs["abssum"])
    assert torch.allclose(f[cs["idx"]], cs["samples"], rtol=1e-7, atol=1e-9)


LINEAR_CASES = ["matrix_notasks", "matrix_tasks", "matrix_xtasks", "matrix_xtasks_nobias_r", "matrixv2_xtasks",
                "matrixv2_tasks", "addition_xtasks", "r0"]


def _run_linear(c, requires_grad=True):
    P = {k: v.clone().requires_grad_(requires_grad) for k, v in c["params"].items()}
    x = c["x"].clone().requires_grad_(requires_grad)
    xt = {t: v.clone().requires_grad_(requires_grad) for t, v in c["x_tasks"].items()} if c["x_tasks"] else None
    tasks = c["tasks"]
    has_lora = c["r"]["shared"] > 0
    y, yt = O.mtlora_linear(
        x, P["linear.weight"], P.get("linear.bias"), P.get("lora_shared_A"), P.get("lora_shared_B"), c["scale_s"],
        tasks=tasks if has_lora else None,
        A_t={t: P["lora_tasks_A." + t] for t in tasks} if tasks else None,
        B_t={t: P["lora_tasks_B." + t] for t in tasks} if tasks else None,
        scale_t=c["scale_t"] if tasks else None, x_tasks=xt, shared_mode=c["mode"],
        lora_norm=(P.get("lora_norm.weight"), P.get("lora_norm.bias")))
    return P, x, xt, y, yt


@pytest.mark.parametrize("case", LINEAR_CASES)
def test_linear_forward_backward(golden, case):
    c = golden("linear.pt")[case]
    P, x, xt, y, yt = _run_linear(c)
    close(y, c["y"])
    loss = (y * c["gy"]).sum()
    if c["y_tasks"] is not None:
        for t in c["tasks"]:
            close(yt[t], c["y_tasks"][t])
            loss = loss + (yt[t] * c["gy_tasks"][t]).sum()
    else:
        assert yt is None
    loss.backward()
    close(x.grad, c["dx"])
    if xt is not None:
        for t in c["tasks"]:
            close(xt[t].grad, c["dx_tasks"][t])
    for n, g in c["grads"].items():
        close(P[n].grad, g)


@pytest.mark.parametrize("case", ["matrix_notasks", "matrix_tasks", "matrix_xtasks", "matrix_xtasks_nobias_r"])
def test_linear_closed_form_backward(golden, case):
    """SURVEY §8 a4 formulas == reference autograd."""
    c = golden("linear.pt")[case]
    P = c["params"]
    tasks = c["tasks"]
    x2 = c["x"].reshape(-1, c["x"].shape[-1])
    f = lambda v: v.reshape(-1, v.shape[-1])
    r = O.mtlora_linear_backward_closed_form(
        x2, P["linear.weight"], P["lora_shared_A"], P["lora_shared_B"], c["scale_s"], tasks,
        {t: P["lora_tasks_A." + t] for t in tasks} if tasks else None,
        {t: P["lora_tasks_B." + t] for t in tasks} if tasks else None,
        c["scale_t"] if tasks else None,
        {t: f(v) for t, v in c["x_tasks"].items()} if c["x_tasks"] else None,
        f(c["gy"]), {t: f(v) for t, v in c["gy_tasks"].items()})
    close(r["dx"], f(c["dx"]))
    close(r["dA_s"], c["grads"]["lora_shared_A"])
    close(r["dB_s"], c["grads"]["lora_shared_B"])
    for t in tasks or []:
        close(r[f"dA_t.{t}"], c["grads"]["lora_tasks_A." + t])
        close(r[f"dB_t.{t}"], c["grads"]["lora_tasks_B." + t])
        if c["x_tasks"]:
            close(r["dx_t"][t], f(c["dx_tasks"][t]))


@pytest.mark.parametrize("case", ["sq", "rect", "noshift", "unit"])
def test_window_ops(golden, case):
    c = golden("window_ops.pt")[case]
    B, H, W, C, ws, shift = c["dims"]
    assert torch.equal(O.roll_and_window_partition(c["x"], shift, ws), c["partitioned"])
    assert torch.equal(O.window_merge_and_roll(c["w"], shift, ws, H, W), c["merged"])
    # adjointness (what the *_backward kernels compute): <P x, w> == <x, P^T w>
    back = O.window_merge_and_roll(c["w"], shift, ws, H, W)
    assert torch.allclose((O.roll_and_window_partition(c["x"], shift, ws) * c["w"]).sum(), (c["x"] * back).sum())


@pytest.mark.parametrize("case", ["nomask", "mask"])
def test_window_attention(golden, case):
    c = golden("window_attention.pt")[case]
    assert torch.equal(O.relative_position_index(c["ws"]), c["rel_index"])
    P = {k: v.clone().requires_grad_(True) for k, v in c["params"].items()}
    x = c["x"].clone().requires_grad_(True)
    mt = O.mtlora_config(c["tasks"], r_shared=8, r_task=4, dropout=0.0)
    PP = {"a." + k: v for k, v in P.items()}
    qkv, _ = O._lin(PP, "a.qkv", x, None, None, 0, mt, False, None)
    close(qkv, c["qkv"])
    bias = O.dense_relative_bias(P["relative_position_bias_table"], c["ws"])
    core = O.window_attention_core(qkv, bias, c["mask"], c["heads"])
    y, yt = O._lin(PP, "a.proj", core, None, c["tasks"] if c["lora"] else None, 0, mt, False, None)
    close(y, c["y"])
    loss = (y * O.det_tensor(f"att.{case}.gy", y.shape, 1.0).double()).sum()
    if c["y_tasks"]:
        for t in c["tasks"]:
            close(yt[t], c["y_tasks"][t])
            loss = loss + (yt[t] * O.det_tensor(f"att.{case}.gy.{t}", y.shape, 1.0).double()).sum()
    loss.backward()
    close(x.grad, c["dx"])
    for n, g in c["grads"].items():
        close(P[n].grad, g)


def test_shifted_window_mask_values():
    m = O.shifted_window_mask(14, 14, 7, 3)
    assert m.shape == (4, 49, 49) and set(m.unique().tolist()) == {0.0, -100.0}
    assert O.shifted_window_mask(14, 14, 7, 0) is None


@pytest.mark.parametrize("case", ["shift_lora", "noshift_plain"])
def test_swin_block(golden, case):
    c = golden("swin_block.pt")[case]
    tasks = c["tasks"]
    mt = O.mtlora_config(tasks, r_shared=8, r_task=4, dropout=0.0)
    shapes = {}
    cfg = dict(embed_dim=64, depths=[2], num_heads=[2], window_size=7, tasks=tasks, mtlora=mt, patch_size=4,
               img_size=56)
    full = O.backbone_param_shapes(cfg)
    j = 1 if c["lora"] else 0
    pre = f"layers.0.blocks.{j}."
    P = {}
    for n in c["param_names"]:
        P["b." + n] = torch.empty(full[pre + n])
    O.det_fill_([(k[2:], v) for k, v in P.items()])
    P = {k: v.double().requires_grad_(True) for k, v in P.items()}
    x = c["x"].clone().requires_grad_(True)
    y, yt = O.swin_block(P, "b", x, 14, 14, 2, 7, c["shift"], tasks if c["lora"] else None, 0, mt)
    close(y, c["y"])
    loss = (y * O.det_tensor(f"blk.{case}.gy", y.shape, 1.0).double()).sum()
    if c["y_tasks"]:
        for t in tasks:
            close(yt[t], c["y_tasks"][t])
            loss = loss + (yt[t] * O.det_tensor(f"blk.{case}.gy.{t}", y.shape, 1.0).double()).sum()
    else:
        assert yt is None
    loss.backward()
    close(x.grad, c["dx"])
    for n, g in c["grads"].items():
        if isinstance(g, dict):
            check_sum(P["b." + n].grad, g)
        else:
            close(P["b." + n].grad, g)


def test_losses(golden):
    c = golden("losses.pt")
    for t, d in c.items():
        pred = d["pred"].clone().requires_grad_(True)
        l = O.task_loss(t, pred, d["label"])
        assert abs(l.item() - d["loss"]) < 1e-6, t
        l.backward()
        assert torch.allclose(pred.grad, d["dpred"], rtol=1e-5, atol=1e-8), t


def test_backbone_small(golden):
    c = golden("backbone_small.pt")
    tasks = c["tasks"]
    cfg = O.swin_t_cfg(img_size=56, tasks=tasks, r_shared=8, r_task=4, depths=(2, 2), num_heads=(3, 6),
                       drop_path_rate=0.1, dropout=0.05)
    shapes = O.backbone_param_shapes(cfg)
    ref_names = [n for n in c["names"] if not n.endswith(("attn_mask", "relative_position_index"))]
    assert sorted(shapes) == sorted(ref_names)
    P = {k: v.double().requires_grad_(True) for k, v in O.make_params(shapes).items()}
    x = O.det_tensor("bbs.x", (1, 3, 56, 56), 1.0).double()
    stages = O.backbone_stages(P, x, cfg)
    loss = 0
    for i, (s, tl) in enumerate(stages):
        close(s, c["stages"][i][0], rtol=1e-8, atol=1e-9)
        loss = loss + (s * O.det_tensor(f"bbs.g.{i}", s.shape, 1.0).double()).sum()
        for t in tasks:
            close(tl[t], c["stages"][i][1][t], rtol=1e-8, atol=1e-9)
            loss = loss + (tl[t] * O.det_tensor(f"bbs.g.{i}.{t}", s.shape, 1.0).double()).sum()
    loss.backward()
    for n, g in c["grads"].items():
        if g is None:
            assert P[n].grad is None or P[n].grad.abs().max() == 0
        elif isinstance(g, dict):
            check_sum(P[n].grad, g, rtol=1e-7)
        else:
            close(P[n].grad, g, rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("case", ["downsampler", "intermediate", "trainable_scale"])
def test_backbone_options(golden, case):
    """DOWNSAMPLER_ENABLED / INTERMEDIATE_SPECIALIZATION / TRAINABLE_SCALE_SHARED backbones vs the reference fixture."""
    c = golden("backbone_options.pt")[case]
    tasks = c["tasks"]
    cfg = O.swin_t_cfg(img_size=56, tasks=tasks, r_shared=8, r_task=4, depths=(2, 2), num_heads=(3, 6),
                       drop_path_rate=0.1, dropout=0.05, **c["over"])
    shapes = O.backbone_param_shapes(cfg)
    assert sorted(shapes) == sorted(n for n in c["names"] if not n.endswith(("attn_mask", "relative_position_index")))
    P = {k: v.double() for k, v in O.make_params(shapes).items()}
    for n in c["trainable"]:
        P[n].requires_grad_(True)
    assert sorted(n for n in P if O.trainable_filter("backbone." + n)) == sorted(c["trainable"])
    x = O.det_tensor("bbo.x", (1, 3, 56, 56), 1.0).double()
    stages = O.backbone_stages(P, x, cfg)
    loss = 0
    for i, (s, tl) in enumerate(stages):
        close(s, c["stages"][i][0], rtol=1e-8, atol=1e-9)
        loss = loss + (s * O.det_tensor(f"bbo.g.{i}", s.shape, 1.0).double()).sum()
        for t in tasks:
            close(tl[t], c["stages"][i][1][t], rtol=1e-8, atol=1e-9)
            loss = loss + (tl[t] * O.det_tensor(f"bbo.g.{i}.{t}", s.shape, 1.0).double()).sum()
    loss.backward()
    for n, g in c["grads"].items():
        if isinstance(g, dict):
            check_sum(P[n].grad, g, rtol=1e-7)
        else:
            close(P[n].grad, g, rtol=1e-7, atol=1e-8)
    assert sorted(n for n in c["trainable"] if P[n].grad is None or P[n].grad.abs().max() == 0) == c["grad_is_none"]


def test_c2_structure(golden):
    """state-dict names/shapes and the trainable set of the C2 model."""
    c = golden("c2_structure.pt")
    tasks = ["semseg", "normals", "sal", "human_parts"]
    cfg = O.swin_t_cfg(448, tasks, 64, 4)
    shapes = {("backbone." + k): v for k, v in O.backbone_param_shapes(cfg).items()}
    shapes.update(O.head_param_shapes(cfg, O.NUM_OUTPUT))
    ref = {k: v for k, v in c["state"].items()
           if not k.endswith(("attn_mask", "relative_position_index", "num_batches_tracked"))}
    assert shapes == ref
    assert len(c["state"]) == 431 and c["n_params"] == 34262906 and c["n_trainable"] == 8344634
    params = [k for k in shapes if not k.endswith(("running_mean", "running_var"))]
    mine = sorted(k for k in params if O.trainable_filter(k))
    assert mine == sorted(c["trainable"])


@pytest.mark.timeout(600)
def test_c1_model(golden):
    """BASELINE configs[0]: Swin-T/224, 1 task, r=4, bs=2, CPU fwd+bwd."""
    c = golden("c1_model.pt")
    torch.set_num_threads(8)
    tasks = ["semseg"]
    cfg = O.swin_t_cfg(224, tasks, 4, 4, drop_path_rate=0.2)
    shapes = {("backbone." + k): v for k, v in O.backbone_param_shapes(cfg).items()}
    shapes.update(O.head_param_shapes(cfg, {"semseg": 21}))
    P = O.make_params(shapes)
    n_params = sum(v.numel() for k, v in P.items() if not k.endswith(("running_mean", "running_var")))
    assert n_params == c["n_params"] == 28370271
    for k, v in P.items():
        if O.trainable_filter(k) and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    assert sum(v.numel() for v in P.values() if v.requires_grad) == c["n_trainable"] == 2451999
    img, tg = O.synthetic_batch(2, 224, tasks, seed=1234)
    out = O.full_model(P, img, cfg, train=False)
    loss, _ = O.multi_task_loss(out, tg, tasks)
    assert abs(loss.item() - c["loss"]) < 2e-5 * abs(c["loss"]), (loss.item(), c["loss"])
    check_sum(out["semseg"], c["out"], rtol=2e-5)
    loss.backward()
    for n, cs in c["grads"].items():
        g = P[n].grad.double().flatten()
        ref = cs["samples"]
        assert torch.allclose(g[cs["idx"]], ref, rtol=2e-3, atol=2e-6 * max(1e-12, ref.abs().max().item()) + 1e-9), n
    none = sorted(k for k, v in P.items() if v.requires_grad and v.grad is None)
    assert none == c["grad_is_none"]


def test_dropout_mask_torch_restatement_equals_numpy_form():
    """oracle.dropout_keep_mask_t (torch int64, any device, row offset) == oracle.dropout_keep_mask (numpy uint64), the
    definition mirrored from csrc/common.h: seeds with high bits set, odd sizes, a row offset."""
    for seed, stream, rows, cols, p in [(0x9E3779B97F4A7C15, 0, 257, 96, 0.25), (12345, 3, 64, 384, 0.05),
                                        (0xFFFFFFFFFFFFFFFF, 1, 33, 7, 0.5), (1 << 32, 0, 5, 1536, 0.999)]:
        a = O.dropout_keep_mask(seed, stream, rows, cols, p)
        b = O.dropout_keep_mask_t(seed, stream, rows, cols, p)
        assert torch.equal(a, b), (seed, stream, rows, cols, p)
        c = O.dropout_keep_mask_t(seed, stream, rows - rows // 2, cols, p, row0=rows // 2)
        assert torch.equal(a[rows // 2:], c)
    assert O.dropout_keep_mask_t(1, 0, 4, 8, 0.0).all()
