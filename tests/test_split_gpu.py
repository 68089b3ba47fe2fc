"""Split-fp16 contraction engine (csrc/split_engine.hpp, include/madeleine_amd.h): image construction and the NT / TN products
through the C ABI against fp64, at the tolerance of the exact-fp32 kernels (1e-6 relative, tests/test_hip_kernels.py:test_linear_vs_torch)
-- the engine replaces fp32 contractions of the reference (nn.Linear, madeleine/models/Model.py:351) and must not cost accuracy."""
import numpy as np
import pytest
import torch

from tests._util import rel_err, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _decode(img, rows, K):
    """fp64 value of every element of a split image: (hi + lo) / scale."""
    raw = img.data[:rows].contiguous().view(torch.int16).view(rows, K // 32, 2, 32)
    hl = raw.view(torch.float16).double()
    return (hl[:, :, 0] + hl[:, :, 1]).reshape(rows, K) / float(img.scale[0])


@pytest.mark.parametrize("rows,K,spread", [(300, 512, 1.0), (77, 2048, 1e4), (1, 32, 1.0)])
def test_split_image_represents_fp32(dev, rows, K, spread):
    """hi + lo carries 21+ bits of every value that lies within 2^16 of the tensor maximum (relative error <= max(2^-23,
    2^-25 / |scaled value|)); the scale is a power of two with the scaled maximum in [2^13, 2^14); pad rows are zero."""
    from madeleine_amd import functional as MF
    x = t((rows, K), f"spi:{rows}{K}") * torch.logspace(0, np.log10(spread), K)
    img = MF.split_image(x.to(dev), pad_rows=32)
    s, amax = float(img.scale[0]), float(img.scale[1])
    assert amax == float(x.abs().max()) and np.log2(s) == int(np.log2(s)) and 2 ** 13 <= amax * s < 2 ** 14
    dec = _decode(img, rows, K).cpu()
    big = x.abs() >= amax * 2.0 ** -16
    assert float(((dec - x.double()).abs() / x.double().abs().clamp_min(1e-300))[big].max()) < 2.0 ** -21
    small = ~big                                                               # graceful below: absolute error <= 2^-38 of the maximum
    assert (not small.any()) or float((dec - x.double()).abs()[small].max()) <= amax * 2.0 ** -38
    assert float(img.data[rows:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(300, 512, 512), (1000, 2048, 512), (513, 800, 512), (2500, 512, 1056), (257, 4, 32),
                                   (256 * 3, 256, 2048),
                                   # N <= 128 with more than 256 rows: the 512 x 128 tile (the token_projector's shape, ragged rows / columns)
                                   (3000, 128, 2048), (513, 128, 512), (1024, 96, 64), (70000, 128, 512), (511, 4, 32)])
def test_split_gemm_nt_vs_fp64(dev, M, N, K):
    from madeleine_amd import functional as MF
    a = t((M, K), f"spn:a{M}{K}") * 3
    b = 0.05 * t((N, K), f"spn:b{N}{K}")
    bias = 0.3 * t((N,), f"spn:c{N}")
    ref = a.double() @ b.double().t()
    A, B = MF.split_image(a.to(dev)), MF.split_image(b.to(dev))
    C = MF.split_gemm_nt(A, B)
    assert rel_err(C, ref) < 1e-6
    amax = torch.zeros(1, device=dev)
    C2 = MF.split_gemm_nt(A, B, bias.to(dev), out=C.clone(), accumulate=True, absmax_out=amax)
    assert rel_err(C2, 2 * ref + bias.double()) < 1e-6
    assert float(amax) == float(C2.abs().max())


def test_split_gemm_nt_error_not_above_fp32_chain(dev):
    """Elementwise, relative to sum_k |a_k b_k|: the 3-term split product is at least as accurate as an fp32 fmaf accumulation
    (here: the exact-fp32 matrix-core kernel of the 'fp32' GEMM mode)."""
    from madeleine_amd import functional as MF
    M, N, K = 1024, 512, 2048
    a, b = t((M, K), "spe:a") * 2, t((N, K), "spe:b") / 45.0
    ref = a.double() @ b.double().t()
    mag = a.double().abs() @ b.double().abs().t()
    C = MF.split_gemm_nt(MF.split_image(a.to(dev)), MF.split_image(b.to(dev))).cpu().double()
    C32 = MF.LinearFn.apply(a.to(dev), b.to(dev), None).cpu().double()
    e_split, e_f32 = ((C - ref).abs() / mag), ((C32 - ref).abs() / mag)
    assert float(e_split.max()) < 4e-7 and float(e_split.pow(2).mean().sqrt()) <= 1.5 * float(e_f32.pow(2).mean().sqrt())


@pytest.mark.parametrize("T,Mi,N", [(1000, 512, 512), (12325, 512, 2048), (4129, 800, 512), (33, 32, 32), (70000, 512, 1024)])
def test_split_gemm_tn_vs_fp64(dev, T, Mi, N):
    """dW-type product over the token rows of two images: ragged token tail, several token splits, ragged column tiles."""
    from madeleine_amd import functional as MF
    x = t((T, Mi), f"spt:x{T}{Mi}") * 2
    dy = t((T, N), f"spt:d{T}{N}") * torch.logspace(0, -3, T).unsqueeze(1)        # tokens with 1000x smaller gradients
    ref = dy.double().t() @ x.double()
    out = MF.split_gemm_tn(MF.split_image(x.to(dev)), MF.split_image(dy.to(dev), pad_rows=32))
    assert tuple(out.shape) == (N, Mi)
    assert rel_err(out, ref) < 1e-6


def test_split_linear_autograd_matches_fp32_mode(dev):
    """functional.linear in the two GEMM modes: same values / gradients (1e-6 against fp64), bias path included."""
    from madeleine_amd import functional as MF
    T, N, K = 3000, 512, 800
    x, W, b, dy = t((T, K), "spl:x"), 0.05 * t((N, K), "spl:w"), 0.3 * t((N,), "spl:b"), t((T, N), "spl:dy")
    x64, W64, b64 = (v.double().requires_grad_() for v in (x, W, b))
    (x64 @ W64.t() + b64).backward(dy.double())
    old = MF.gemm_mode()
    try:
        for mode in ("split", "fp32"):
            MF.set_gemm_mode(mode)
            xd, Wd, bd = (v.to(dev).requires_grad_() for v in (x, W, b))
            y = MF.linear(xd, Wd, bd)
            y.backward(dy.to(dev))
            assert rel_err(y, (x64 @ W64.t() + b64).detach()) < 1e-6, mode
            assert rel_err(xd.grad, x64.grad) < 1e-6 and rel_err(Wd.grad, W64.grad) < 1e-6 and rel_err(bd.grad, b64.grad) < 1e-6, mode
    finally:
        MF.set_gemm_mode(old)


def test_parity_suite_in_fp32_gemm_mode(dev):
    """The default GEMM mode is 'split' (what every other -m gpu test exercises).  The exact-fp32 matrix-core kernels
    (v_mfma_f32_32x32x2_f32, MADELEINE_GEMM=fp32) stay in the product as the second mode: the encoder goldens with every parameter
    gradient, the gate kernels on the multi-split path against the CPU oracle, the full train step golden and the ragged config-5
    backward run here once more in that mode."""
    from madeleine_amd import functional as MF
    from tests import test_bench_path_gpu as TB
    from tests import test_model_gpu as TM
    old = MF.gemm_mode()
    MF.set_gemm_mode("fp32")
    try:
        TM.test_encoder_eval_and_grads(dev)
        TM.test_stain_encoding_quirk(dev)          # round 6: the stain fold through the grouped LayerNorm pass of this engine
        TB.test_gate_split_path_vs_oracle(dev, 0.25)
        TB.test_full_step_dp_w1_matches_reference_golden(dev)
        TM.test_forward_ragged_unequal_backward_vs_oracle(dev, False)
    finally:
        MF.set_gemm_mode(old)


def test_gemm_mode_switch():
    from madeleine_amd import functional as MF
    old = MF.gemm_mode()
    try:
        MF.set_gemm_mode("fp32")
        assert MF.gemm_mode() == "fp32"
        with pytest.raises(ValueError):
            MF.set_gemm_mode("bf16")
    finally:
        MF.set_gemm_mode(old)


@pytest.mark.parametrize("T,K,N,want_fp32,mode", [(1000, 512, 512, False, "eval"), (700, 800, 512, False, "mask"), (515, 512, 2048, True, "mask"),
                                                   (300, 64, 1024, True, "eval")])
def test_preattn_block_vs_torch_fp64(dev, T, K, N, want_fp32, mode):
    """One pre-attention block (Linear -> LayerNorm -> GELU -> Dropout, reference madeleine/models/Model.py:351-354) as the fused split
    node: the image it emits decodes to the fp64 result, the optional fp32 copy equals it, and x / W / bias / gamma / beta gradients
    match fp64 autograd -- the LayerNorm kernels write split images straight away (forward: scale from the parameter bound; backward:
    from rstd_max, max|dy|), so this also checks those bounds leave the values well inside the fp16 range."""
    import torch.nn.functional as F
    from madeleine_amd import functional as MF
    from oracle import recipe
    x = t((T, K), f"pb:x{T}{K}") * 2
    W = 0.05 * t((N, K), f"pb:w{N}{K}")
    lb, g, b = 0.3 * t((N,), f"pb:lb{N}"), 1 + 0.2 * t((N,), f"pb:g{N}"), 0.3 * t((N,), f"pb:b{N}")
    dy = t((T, N), f"pb:dy{T}{N}") * torch.logspace(0, -2, T).unsqueeze(1)
    keep = torch.from_numpy(recipe.bernoulli((T, N), f"pb:k{T}{N}", 0.9)) if mode == "mask" else None
    p = 0.1 if keep is not None else 0.0
    leaves = [v.double().requires_grad_() for v in (x, W, lb, g, b)]
    ref = F.gelu(F.layer_norm(leaves[0] @ leaves[1].t() + leaves[2], (N,), leaves[3], leaves[4], 1e-5))
    if keep is not None:
        ref = ref * keep.double() / 0.9
    ref.backward(dy.double())
    dl = [v.to(dev).requires_grad_() for v in (x, W, lb, g, b)]
    img, sc, out = MF.preattn_block(dl[0], None, dl[1], dl[2], dl[3], dl[4], 1e-5, p, 0,
                                    None if keep is None else keep.to(torch.uint8).to(dev), want_fp32)
    dec = _decode(MF.SplitImage(img.detach(), sc, T, N), T, N)
    assert rel_err(dec, ref) < 1e-5
    bound = float(sc[1])
    assert float(ref.abs().max()) <= bound < 2 ** 9 * float(ref.abs().max()) and 2 ** 13 <= bound * float(sc[0]) < 2 ** 14
    if want_fp32:
        assert rel_err(out, ref) < 1e-5
        out.backward(dy.to(dev))
    else:
        img.backward(dy.to(dev))
    for name, a, r in zip(("x", "W", "lin_bias", "gamma", "beta"), dl, leaves):
        assert rel_err(a.grad, r.grad) < 2e-5, name


def test_preattn_blocks_chain_on_images(dev):
    """Two chained blocks: the second consumes the first's IMAGE (no fp32 activation in between); gradients flow back as fp32."""
    import torch.nn.functional as F
    from madeleine_amd import functional as MF
    T, K, N1, N2 = 900, 512, 512, 2048
    x = t((T, K), "pbc:x")
    W1, W2 = 0.05 * t((N1, K), "pbc:w1"), 0.05 * t((N2, N1), "pbc:w2")
    g1, b1, g2, b2 = 1 + 0.1 * t((N1,), "pbc:g1"), 0.1 * t((N1,), "pbc:b1"), 1 + 0.1 * t((N2,), "pbc:g2"), 0.1 * t((N2,), "pbc:b2")
    dy = t((T, N2), "pbc:dy")
    lv = [v.double().requires_grad_() for v in (x, W1, g1, b1, W2, g2, b2)]
    h = F.gelu(F.layer_norm(lv[0] @ lv[1].t(), (N1,), lv[2], lv[3], 1e-5))
    ref = F.gelu(F.layer_norm(h @ lv[4].t(), (N2,), lv[5], lv[6], 1e-5))
    ref.backward(dy.double())
    dl = [v.to(dev).requires_grad_() for v in (x, W1, g1, b1, W2, g2, b2)]
    i1, s1, _ = MF.preattn_block(dl[0], None, dl[1], None, dl[2], dl[3])
    i2, s2, out = MF.preattn_block(i1, s1, dl[4], None, dl[5], dl[6], want_fp32=True)
    assert rel_err(out, ref) < 1e-5
    out.backward(dy.to(dev))
    for name, a, r in zip(("x", "W1", "g1", "b1", "W2", "g2", "b2"), dl, lv):
        assert rel_err(a.grad, r.grad) < 2e-5, name


@pytest.mark.parametrize("in_scale,loss_scale", [(1.0, 1.0), (3e3, 1e-7), (1e-3, 1e6)])
def test_split_mode_is_scale_robust(dev, in_scale, loss_scale):
    """fp16 has five exponent bits: the split engine relies on per-tensor power-of-two scales (exact absmax for inputs and weights,
    rigorous bounds for the LayerNorm / dz outputs).  A whole encoder + pooling + projector forward / backward with the inputs and the
    loss scaled over ten orders of magnitude gives the same slide embeddings and parameter gradients as the exact-fp32 matrix-core mode."""
    from types import SimpleNamespace
    from madeleine_amd import MADELEINE
    from madeleine_amd import functional as MF
    from oracle import recipe
    mods = ["HE", "HER2", "ER"]
    cfg = SimpleNamespace(MODALITIES=mods, wsi_encoder="abmil", patch_embedding_dim=64, wsi_encoder_hidden_dim=512, activation="softmax",
                          n_heads=4)
    model = MADELEINE(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.state_dict_recipe(shapes, "scl").items()})
    model = model.to(dev).eval()
    feats = t((3, 3, 200, 64), "scl:feats") * in_scale
    w_e, w_t = t((3, 1, 512), "scl:we").to(dev), t((3, 200, 128), "scl:wt").to(dev)
    res = {}
    old = MF.gemm_mode()
    try:
        for mode in ("fp32", "split"):
            MF.set_gemm_mode(mode)
            model.zero_grad()
            embs, toks = model({"feats": feats}, device=dev, train=True)
            obj = loss_scale * (sum((embs[k] * (w_e if k != "HE" else w_e.unsqueeze(3))).sum() for k in mods)
                                + 0.01 * sum((toks[k] * (w_t if k != "HE" else w_t.unsqueeze(3))).sum() for k in mods))
            obj.backward()
            res[mode] = (embs["ER"].detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
    finally:
        MF.set_gemm_mode(old)
    assert rel_err(res["split"][0], res["fp32"][0]) < 1e-5
    top = max(float(g.norm()) for g in res["fp32"][1].values())
    assert np.isfinite(top) and top > 0
    for k, g in res["fp32"][1].items():
        assert torch.isfinite(res["split"][1][k]).all(), k
        assert float((res["split"][1][k] - g).norm()) <= 1e-4 * float(g.norm()) + 1e-6 * top, k


def test_split_gemm_nt_row_gate(dev):
    """Accumulate-mode product with the row gate: tiles of 256 all-zero A rows are skipped and the result equals the ungated one."""
    from madeleine_amd import functional as MF
    M, N, K = 1300, 2048, 128
    a = t((M, K), "spg:a")
    a[100:1000] = 0                      # tiles 1 and 2 (rows 256..767) are entirely zero, tiles 0 and 3 partly
    b = 0.05 * t((N, K), "spg:b")
    c0 = t((M, N), "spg:c")
    A, B = MF.split_image(a.to(dev)), MF.split_image(b.to(dev))
    gate = MF.split_tile_absmax(a.to(dev))
    assert gate.cpu().tolist()[1:3] == [0.0, 0.0] and float(gate[0]) > 0 and float(gate[5]) > 0
    out_g = MF.split_gemm_nt(A, B, out=c0.to(dev).clone(), accumulate=True, row_gate=gate)
    out_u = MF.split_gemm_nt(A, B, out=c0.to(dev).clone(), accumulate=True)
    assert torch.equal(out_g, out_u)
    assert rel_err(out_g, c0.double() + a.double() @ b.double().t()) < 1e-6
    with pytest.raises(RuntimeError):
        MF.split_gemm_nt(A, B, out=c0.to(dev).clone(), accumulate=False, row_gate=gate)


def test_split_gemm_tn_chunk_gate(dev):
    """TN product with the chunk list: 32-row chunks in which the B source is identically zero are skipped; the result is unchanged."""
    from madeleine_amd import functional as MF
    T, Mi, N = 9000, 512, 128
    x = t((T, Mi), "spc:x")
    dy = t((T, N), "spc:dy")
    dy[40:5000] = 0
    dy[5100:8990] = 0
    ref = dy.double().t() @ x.double()
    A, B = MF.split_image(x.to(dev)), MF.split_image(dy.to(dev), pad_rows=32)
    gate, cm = MF.split_tile_absmax(dy.to(dev), chunks=True)
    assert torch.equal(cm.cpu(), torch.cat([dy, torch.zeros(-T % 32, N)]).view(-1, 32 * N).abs().amax(1))
    assert torch.equal(gate.cpu(), torch.cat([dy, torch.zeros(-T % 256, N)]).view(-1, 256 * N).abs().amax(1))
    out_g = MF.split_gemm_tn(A, B, b_chunk_max=cm)
    out_u = MF.split_gemm_tn(A, B)
    assert rel_err(out_g, ref) < 1e-6 and rel_err(out_u, ref) < 1e-6
    z = MF.split_gemm_tn(A, MF.split_image(torch.zeros(T, N, device=dev), pad_rows=32),
                         b_chunk_max=MF.split_tile_absmax(torch.zeros(T, N, device=dev), chunks=True)[1])
    assert float(z.abs().max()) == 0.0


def test_pool_from_image_vs_fp32(dev):
    """The pooling kernels on the split image of E (the split GEMM mode never stores an fp32 E): pooled embeddings, statistics and the
    score gradients against the fp32-E kernels, dense and ragged."""
    from madeleine_amd import functional as MF
    H, N, BM = 4, 700, 5
    E = (t((BM * N, H * 512), "pimg:E") * 2.0).to(dev)
    E[17] = 0.0
    scores = (t((BM * N, H), "pimg:s") * 3.0).to(dev)
    dp = t((BM, H * 512), "pimg:dp").to(dev)
    Ei = MF.split_image(E)
    for cu, nb, mx in ((None, BM, N), (torch.tensor([0, 1, 900, 900, 2100, 3500], device=dev), 5, 1400)):
        p0, m0, l0 = MF.pool_fwd_raw(E, scores, nb, N if cu is None else 0, cu, mx)
        p1, m1, l1 = MF.pool_fwd_img_raw(Ei, scores, nb, N if cu is None else 0, cu, mx)
        assert torch.equal(m0, m1) and torch.equal(l0, l1)
        assert rel_err(p1, p0) < 2e-7
        ds0, ds1 = torch.empty_like(scores), torch.empty_like(scores)
        MF.pool_bwd_raw(E, scores, p0, m0, l0, dp[:nb], None, 0, ds0, 0, nb, N if cu is None else 0, cu, mx)
        MF.pool_dscores_img_raw(Ei, scores, p0, m0, l0, dp[:nb], ds1, 0, nb, N if cu is None else 0, cu, mx)
        assert rel_err(ds1, ds0) < 2e-6


def test_embedder_image_only_matches_fp32_tokens(dev):
    """ABMILEmbedder.forward_headmajor(need_tokens=False) keeps E as an image only (LayerNorm kernel writes 4 B per element, pooling reads
    the image): same pooled embeddings, token projections and parameter gradients as the path that also writes the fp32 E."""
    from madeleine_amd.model import ABMILEmbedder
    torch.manual_seed(3)
    emb = ABMILEmbedder(pre_attention_params={"input_dim": 512, "hidden_dim": 512}, attention_params={
        "model": "ABMIL", "params": {"input_dim": 512, "hidden_dim": 512, "dropout": False, "activation": "softmax", "n_heads": 4,
                                     "n_classes": 1}}).to(dev).eval()
    bags = t((3, 600, 512), "pimg:bags").to(dev)
    Wt = (t((128, 2048), "pimg:wt") * 0.02).to(dev).requires_grad_()
    res = []
    for need in (True, False):
        emb.zero_grad()
        Wt.grad = None
        pooled, E, scores, tok = emb.forward_headmajor(bags, tok_proj=(Wt, None), need_tokens=need)
        assert (E is None) == (not need)
        (pooled.square().sum() + tok[:, :40].square().sum()).backward()
        res.append((pooled.detach(), tok.detach(), Wt.grad.clone(), [p.grad.clone() for p in emb.parameters() if p.grad is not None]))
    assert rel_err(res[1][0], res[0][0]) < 1e-6 and rel_err(res[1][1], res[0][1]) < 1e-6
    assert rel_err(res[1][2], res[0][2]) < 1e-5
    top = max(float(b.norm()) for b in res[0][3])
    for a, b in zip(res[1][3], res[0][3]):   # (measured <= 2.7e-5: the two paths pool values that differ in the last bit)
        if b.numel() == 1:   # attention_c.bias: shift invariance of the softmax makes its gradient rounding noise
            assert float((a - b).abs().max()) < 1e-6 * top
        else:
            assert rel_err(a, b) < 1e-4


def test_gate_backward_phase_masks_compose(dev):
    """mdl_abmil_attnpool_bwd_split's phase bit mask (include/madeleine_amd.h): the dz pass (1), then the dX contraction alone (4), then
    the dW contraction alone (8) on the same workspace produce the bits of the one-call form (3 = dz + both contractions); and a
    CU-masked HIP stream can be created and released through the ABI (mdl_stream_create_cu_mask / mdl_stream_destroy)."""
    import ctypes
    from madeleine_amd import _native
    from madeleine_amd import functional as MF
    H, T = 2, 700
    E = t((T, H * 512), "ph:E").to(dev)
    Wa, Wb = (t((H, 512, 512), "ph:Wa") * 0.05).to(dev), (t((H, 512, 512), "ph:Wb") * 0.05).to(dev)
    ba, bb = (t((H, 512), "ph:ba") * 0.1).to(dev), (t((H, 512), "ph:bb") * 0.1).to(dev)
    wc, bc = (t((H, 512), "ph:wc") * 0.1).to(dev), (t((H,), "ph:bc") * 0.1).to(dev)
    ds = t((T, H), "ph:ds").to(dev)
    Ei = MF.split_image(E)
    _s, a, b = MF.gate_fwd_split_raw(Ei, Wa, ba, Wb, bb, wc, bc, 0.25, 1234, None, None, True)
    outs = []
    for phases in (None, (3,), (1, 4, 8), (1, 8, 4)):
        dE = torch.zeros_like(E)
        g = MF.attnpool_bwd_split_raw(Ei, Wa, Wb, wc, a, b, ds, dE, 0.25, 1234, None, None, None, None, None, None, None, 0, phases=phases)
        outs.append((dE,) + tuple(g))
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            assert torch.equal(x, y)
    assert float(outs[0][0].abs().max()) > 0 and float(outs[0][1].abs().max()) > 0
    lib = _native.lib()
    words = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
    stream = ctypes.c_void_p()
    assert lib.mdl_stream_create_cu_mask(8, words, ctypes.byref(stream)) == 0 and stream.value
    assert lib.mdl_stream_destroy(stream) == 0
