"""MADELEINE slide encoder -- MI355X-native mirror of the reference's `MADELEINE`, `ABMILEmbedder`
and `create_model` (reference madeleine/models/Model.py:15-43, :45-216, :314-451).

Same constructor / forward / encode_he contracts, same sub-module names and therefore the same
state_dict keys and shapes (reference and HuggingFace checkpoints load unchanged, with or without a
`module.` prefix).  What differs is where the work happens:

  * pre_attn (3 x Linear + fused LayerNorm-GELU-Dropout), token_projector and projector run in the hand-written
    fp32 matrix-core Linear kernels (functional.linear / ln_gelu_drop) -- in the bf16
    mode (torch.autocast(bfloat16)) in their bf16 siblings (mdl_linear_*_bf16: bf16 activations, fp32 parameters);
  * gated attention scores + softmax-over-patches + weighted pooling, forward and backward, are the
    hand-written HIP kernels behind madeleine_amd.functional.attn_pool.

Head-major trick: the reference interleaves heads as channel j = e*H + c (rearrange
'b t (e c) -> b t e c', Model.py:396).  We permute the ROWS of pre_attn.8 / LayerNorm 9 (and the
COLUMNS of token_projector / projector) by a fixed index vector at forward time, so the encoder emits
the very same numbers with channel order j' = c*512 + e.  Every kernel then reads contiguous 2 KiB
head rows and no activation is ever permuted.  Parameters keep the reference layout.
"""
from collections import OrderedDict
from typing import Dict, Optional, Union

import numpy as np
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import functional as MF
from .abmil import BatchedABMIL, activate

HE_POSITION = 0
PRE_DROPOUT_P = 0.1  # Model.py:354,358,362


class _PermuteFn(torch.autograd.Function):
    """w.index_select(dim, perm) for a BIJECTIVE perm: the backward is the gather by the inverse permutation (one kernel)
    instead of autograd's index_put_(accumulate=True), which sorts the indices (~10 small launches per use and step)."""

    @staticmethod
    def forward(ctx, w, perm, inv, dim):
        ctx.save_for_backward(inv)
        ctx.dim = dim
        return w.index_select(dim, perm)

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return g.index_select(ctx.dim, inv), None, None, None


def bf16_mode() -> bool:
    """True inside torch.autocast(device_type='cuda', dtype=torch.bfloat16): selects the kernels' bf16 mode."""
    return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16


def _strip_module_prefix(state_dict):
    """Model.py:31-40 / utils.py:112-120: DataParallel checkpoints carry a 'module.' prefix."""
    if not any(k.startswith("module.") for k in state_dict):
        return state_dict
    return OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())


def create_model(model_cfg: Union[str, Dict], device: Union[str, torch.device] = 'cpu',
                 checkpoint_path: Optional[str] = None):
    """Mirror of Model.py:15-43."""
    model = MADELEINE(config=model_cfg, stain_encoding=False).to(device)
    if checkpoint_path:
        state_dict = torch.load(checkpoint_path, weights_only=True, map_location=device)   # a state dict: safe unpickler
        model.load_state_dict(_strip_module_prefix(state_dict), strict=True)
        print("* Loaded weights successfully!")
    return model


class ABMILEmbedder(nn.Module):
    """Multi-head gated-ABMIL patch aggregator (Model.py:314-451)."""

    def __init__(self, pre_attention_params: dict = None, attention_params: dict = None, aggregation: str = 'regular') -> None:
        super().__init__()
        self.pre_attention_params = pre_attention_params
        self.attention_params = attention_params
        self.n_heads = attention_params['params']["n_heads"]
        self._build_pre_attention_params(params=pre_attention_params)
        if attention_params is not None:
            self._build_attention_params(attn_model=attention_params['model'], params=attention_params['params'])
        self.agg_type = aggregation
        H = self.n_heads
        hid = pre_attention_params['hidden_dim']
        # perm[j'] = j with j' = c*hid + e (head-major) and j = e*H + c (reference order)
        jp = torch.arange(hid * H)
        self.register_buffer("_perm", (jp % hid) * H + (jp // hid), persistent=False)
        self.register_buffer("_inv_perm", torch.argsort(self._perm), persistent=False)
        self._injected_keep = None  # dict(pre=[3 masks, reference layout], gate=[(ka,kb) per head]) for parity tests

    def _build_pre_attention_params(self, params):
        H = self.n_heads
        self.pre_attn = nn.Sequential(
            nn.Linear(params['input_dim'], params['hidden_dim']),
            nn.LayerNorm(params['hidden_dim']),
            nn.GELU(),
            nn.Dropout(PRE_DROPOUT_P),
            nn.Linear(params['hidden_dim'], params['hidden_dim']),
            nn.LayerNorm(params['hidden_dim']),
            nn.GELU(),
            nn.Dropout(PRE_DROPOUT_P),
            nn.Linear(params['hidden_dim'], params['hidden_dim'] * H),
            nn.LayerNorm(params['hidden_dim'] * H),
            nn.GELU(),
            nn.Dropout(PRE_DROPOUT_P),
        )

    def _build_attention_params(self, attn_model='ABMIL', params=None):
        if attn_model == 'ABMIL':
            self.attn = nn.ModuleList([BatchedABMIL(**params) for _ in range(self.n_heads)])
        else:
            raise NotImplementedError('Attention model not implemented -- Options are ABMIL')

    # ------------------------------------------------------------------ internals (head-major)
    def permuted(self, w, dim):
        """Parameter `w` with axis `dim` (length H*512) in head-major order."""
        return _PermuteFn.apply(w, self._perm, self._inv_perm, dim)

    def _drop_cfg(self, blk, perm=None):
        """(p, seed, keep) of block `blk`'s dropout for this forward."""
        p, seed, keep = 0.0, 0, None
        if self.training:
            p = float(self.pre_attn[4 * blk + 3].p)   # the nn.Dropout module of this block (0.1 unless the user changed it)
            inj = self._injected_keep
            if p == 0.0:
                pass
            elif inj is not None:
                keep = inj["pre"][blk]
                if perm is not None:
                    keep = keep[..., perm]
                keep = keep.to(torch.uint8).contiguous()
            else:
                seed = MF.new_dropout_seed()
        return p, seed, keep

    def _block_split(self, x, x_scale, blk, perm=None, want_fp32=False, weight=None, group_bias=None):
        """Linear + LayerNorm + GELU + Dropout of block `blk` as one node on the split engine (functional.PreAttnBlockFn)."""
        lin, ln = self.pre_attn[4 * blk], self.pre_attn[4 * blk + 1]
        W, lb, g, b = (lin.weight if weight is None else weight), lin.bias, ln.weight, ln.bias
        if perm is not None:
            W, lb, g, b = self.permuted(W, 0), self.permuted(lb, 0), self.permuted(g, 0), self.permuted(b, 0)
        p, seed, keep = self._drop_cfg(blk, perm)
        if keep is not None:
            keep = keep.reshape(-1, keep.shape[-1])
        gb, rg, cu = group_bias if group_bias is not None else (None, None, None)
        return MF.preattn_block(x, x_scale, W, lb, g, b, ln.eps, p, seed, keep, want_fp32, gb, rg, cu)

    def _act(self, x, ln, blk, perm=None, lin_bias=None):
        """(+ bias of the preceding Linear) -> LayerNorm -> GELU -> Dropout(.1) of block `blk` in ONE fused HIP pass each
        way (functional.ln_gelu_drop); the Linear itself runs bias-free."""
        g, b = (ln.weight, ln.bias) if perm is None else (self.permuted(ln.weight, 0), self.permuted(ln.bias, 0))
        if lin_bias is not None and perm is not None:
            lin_bias = self.permuted(lin_bias, 0)
        p, seed, keep = self._drop_cfg(blk, perm)
        return MF.ln_gelu_drop(x if x.dtype == torch.bfloat16 else x.float(), g, b, ln.eps, p, seed, keep, lin_bias)

    def embed_tokens_headmajor(self, bags: torch.Tensor, return_image: bool = False, want_fp32: bool = True, stain=None):
        """pre_attn(bags) with the 2048 output channels in head-major order: [BM, N, H*512]  (with return_image: (E, image of E or
        None) -- in the split GEMM mode the last block's kernel writes the split image the gate contractions read).  want_fp32 = False
        (with return_image, split GEMM mode only): no fp32 copy of E is written -- the returned "E" IS the image tensor (an opaque
        float32 [BM, N, H*512] tensor that only functional.attn_pool(..., e_only_image=True) can read; `image_only(E, e_img)` tells).

        Under torch.autocast(bfloat16) -- the reference's `precision: bfloat16` runs (trainer.py:101-103) -- the
        activations are kept in bf16 end to end (Linear outputs, the fused LayerNorm-GELU-Dropout input/output and E):
        the bf16 mode of the HIP kernels; LayerNorm statistics, GELU and every reduction stay fp32 inside the kernels."""
        pa = self.pre_attn
        perm = self._perm
        # stain = (e [G, 32], row_group int32 [T], cu_groups int64 [G + 1]): MADELEINE's stain encoding concatenates ONE 32-vector per bag to
        # every patch feature in front of the first Linear (Model.py:125-132, :351).  [x | e_g] W^T = x Wx^T + e_g We^T: on the split engine the
        # concat never exists -- the bag's row e_g We^T ([G, 512], a small HIP Linear with autograd) enters the first product as a per-group
        # bias (mdl_split_gemm_nt_group_bias) and its gradient comes back from the grouped LayerNorm backward.  The other engines (bf16, exact
        # fp32) add the row in a grouped LayerNorm-GELU-Dropout pass (round 6); only more than 2048 bags per call still concatenate.
        group_bias = None
        if stain is not None:
            e_rows, row_group, cu_groups = stain
            d_feat = bags.shape[-1]
            x_flat = bags.reshape(-1, d_feat)
            if (not bf16_mode()) and MF.preattn_split_supported(x_flat, d_feat) and pa[0].weight.shape[0] % 32 == 0 \
                    and e_rows.shape[1] % 4 == 0 and e_rows.shape[0] <= 2048:
                Wfull = pa[0].weight
                img, sc, _ = self._block_split(x_flat.float().contiguous(), None, 0, weight=Wfull[:, :d_feat],
                                               group_bias=(MF.linear(e_rows.float(), Wfull[:, d_feat:].contiguous()), row_group, cu_groups))
                img, sc, _ = self._block_split(img, sc, 1)
                keep_fp32 = want_fp32 or not return_image
                img, sc, E = self._block_split(img, sc, 2, perm, want_fp32=keep_fp32)
                E = (E if keep_fp32 else img).view(*bags.shape[:-1], img.shape[-1])
                return (E, (img, sc)) if return_image else E
            Wfull = pa[0].weight
            if cu_groups is not None and Wfull.shape[0] in (256, 512, 1024) and e_rows.shape[0] <= 2048 \
                    and not os.environ.get("MADELEINE_STAIN_CONCAT"):      # (A/B switch: the round-5 concatenation)
                # bf16 / exact-fp32 engines (round 6): the same fold with the bag's row as a per-group bias of the fused
                # LayerNorm-GELU-Dropout pass (functional.LNGeluDropGroupsFn) -- their GEMM epilogues add no bias
                lowp = bf16_mode()          # (read before autocast is switched off for the kernels' own dtype handling)
                with torch.autocast(device_type="cuda", enabled=False):
                    gb = MF.linear(e_rows.float(), Wfull[:, d_feat:].contiguous()) + pa[0].bias          # [G, 512]
                    Wx = Wfull[:, :d_feat]
                    xin = x_flat
                    if d_feat % 32:
                        padc = 32 - d_feat % 32
                        xin, Wx = F.pad(xin, (0, padc)), F.pad(Wx, (0, padc))
                    xin = xin.to(torch.bfloat16) if lowp else xin.float()
                    p, seed, keep = self._drop_cfg(0, None)
                    if keep is not None:
                        keep = keep.reshape(-1, keep.shape[-1])
                    x = MF.ln_gelu_drop_groups(MF.linear(xin, Wx.contiguous()), pa[1].weight, pa[1].bias, pa[1].eps, p, seed, keep, gb,
                                               cu_groups)
                    x = self._act(MF.linear(x, pa[4].weight), pa[5], 1, None, pa[4].bias)
                    E = self._act(MF.linear(x, self.permuted(pa[8].weight, 0)), pa[9], 2, perm, pa[8].bias)
                E = E.view(*bags.shape[:-1], E.shape[-1])
                return (E, None) if return_image else E
            enc = e_rows.index_select(0, row_group.long()).view(*bags.shape[:-1], e_rows.shape[1])
            bags = torch.cat([bags, enc.to(bags.dtype)], dim=-1)
        # Any patch_embedding_dim (Model.py:351 is a plain nn.Linear): the kernels contract over 32-column blocks, so an input width that
        # is not a multiple of 32 is zero-padded -- features and the columns of the first weight alike, which leaves every product
        # unchanged (exact) and costs one padded copy of the bags per forward; autograd slices the weight gradient back.  The usual
        # encoders (512 / 768 / 1024 / 1536 / 2048 / 2560, + 32 stain channels) never take this branch.
        W0 = pa[0].weight
        k_in = bags.shape[-1]
        if k_in % 32:
            padc = 32 - k_in % 32
            bags = F.pad(bags, (0, padc))
            W0 = F.pad(W0, (0, padc))
        if bf16_mode():
            with torch.autocast(device_type="cuda", enabled=False):
                bf = torch.bfloat16
                x = self._act(MF.linear(bags.to(bf), W0), pa[1], 0, None, pa[0].bias)
                x = self._act(MF.linear(x, pa[4].weight), pa[5], 1, None, pa[4].bias)
                E = self._act(MF.linear(x, self.permuted(pa[8].weight, 0)), pa[9], 2, perm, pa[8].bias)
                return (E, None) if return_image else E
        x2d = bags.reshape(-1, bags.shape[-1])
        if MF.preattn_split_supported(x2d, x2d.shape[-1]) and pa[0].weight.shape[0] % 32 == 0:
            # split GEMM mode: three fused blocks; the activations between them exist as split images only
            img, sc, _ = self._block_split(x2d.float().contiguous(), None, 0, weight=W0)
            img, sc, _ = self._block_split(img, sc, 1)
            keep_fp32 = want_fp32 or not return_image
            img, sc, E = self._block_split(img, sc, 2, perm, want_fp32=keep_fp32)
            E = (E if keep_fp32 else img).view(*bags.shape[:-1], img.shape[-1])
            return (E, (img, sc)) if return_image else E
        x = self._act(MF.linear(bags.float(), W0), pa[1], 0, None, pa[0].bias)
        x = self._act(MF.linear(x, pa[4].weight), pa[5], 1, None, pa[4].bias)
        E = self._act(MF.linear(x, self.permuted(pa[8].weight, 0)), pa[9], 2, perm, pa[8].bias)
        return (E, None) if return_image else E

    @staticmethod
    def image_only(E, e_img) -> bool:
        """True when `E` of embed_tokens_headmajor(..., want_fp32=False) is the image tensor itself."""
        return e_img is not None and E.data_ptr() == e_img[0].data_ptr()

    def gate_params_stacked(self):
        ps = [h.gate_params() for h in self.attn]
        return tuple(torch.stack([p[i] for p in ps]) for i in range(5)) + (torch.cat([p[5] for p in ps]),)

    def _gate_dropout(self, shape_bn):
        """(p, seed, keep_a, keep_b) for this forward; masks are uint8 [T,H,512]."""
        p = self.attn[0].dropout_p() if len(self.attn) else 0.0
        if p == 0.0:
            return 0.0, 0, None, None
        inj = self._injected_keep
        if inj is not None:
            ka = torch.stack([g[0] for g in inj["gate"]], dim=2)  # [BM,N,H,512]
            kb = torch.stack([g[1] for g in inj["gate"]], dim=2)
            T = shape_bn[0] * shape_bn[1]
            return p, 0, ka.reshape(T, self.n_heads, MF.HID).to(torch.uint8).contiguous(), \
                kb.reshape(T, self.n_heads, MF.HID).to(torch.uint8).contiguous()
        return p, MF.new_dropout_seed(), None, None

    def pool_headmajor(self, E_hm: torch.Tensor, views=(), tok_proj=None, e_img=None):
        """E_hm [BM,N,H*512] -> (pooled_hm [BM,(1+V,)H*512], raw scores [BM,N,H]) through the fused HIP path; `views` = V int32
        token-index lists pooled in the same autograd node (no index_select copies of E).  tok_proj = (W [P,H*512] head-major
        columns, bias): the token projection [BM,N,P] comes out of the same node (third result), see functional.AttnPoolFn."""
        for h in self.attn:
            h._check_geometry()
        BM, N, _ = E_hm.shape
        wa, ba, wb, bb, wc, bc = self.gate_params_stacked()
        p, seed, ka, kb = self._gate_dropout((BM, N))
        out = MF.attn_pool(E_hm, wa, ba, wb, bb, wc, bc, p, seed, ka, kb, views=views, tok_proj=tok_proj, e_img=e_img,
                           e_only_image=self.image_only(E_hm, e_img))
        if tok_proj is None:
            return out[0], out[1].view(BM, N, self.n_heads)
        return out[0], out[1].view(BM, N, self.n_heads), out[2].view(BM, N, -1)

    def pool_headmajor_ragged(self, E_hm: torch.Tensor, cu_seqlens: torch.Tensor, max_len: int, e_img=None, tok_proj=None):
        """Packed E_hm [T,H*512] + cu_seqlens int64 [n_bags+1] -> (pooled_hm [n_bags,H*512], raw scores [T,H]) (+ the token projections
        [T,P] with tok_proj = (W, bias), see pool_headmajor)."""
        for h in self.attn:
            h._check_geometry()
        wa, ba, wb, bb, wc, bc = self.gate_params_stacked()
        p, seed, ka, kb = self._gate_dropout((E_hm.shape[0], 1))
        act = self.attn[0].activation
        if act != 'softmax':
            # relu / leaky_relu / sigmoid attention (abmil.py:56-61) on ragged bags: element-wise weights, no normalisation over the patch
            # axis -- raw scores from the gate kernel, the activation in torch, un-normalised weighted pooling over cu_seqlens
            # (mdl_abmil_wpool_*), as the dense path does (forward_headmajor_from_tokens)
            if self.image_only(E_hm, e_img) or tok_proj is not None:
                raise ValueError("activation=%r pools the token embeddings themselves: call embed_tokens_headmajor(want_fp32=True) and project "
                                 "the tokens separately" % act)
            scores = MF.gate_scores(E_hm, wa, ba, wb, bb, wc, bc, p, seed, ka, kb)          # [T, H]
            return MF.weighted_pool(E_hm, activate(scores, act), cu_seqlens, max_len), scores
        return MF.attn_pool(E_hm, wa, ba, wb, bb, wc, bc, p, seed, ka, kb, cu_seqlens, max_len, e_img=e_img, tok_proj=tok_proj,
                            e_only_image=self.image_only(E_hm, e_img))

    def _scores_only(self, E_hm):
        BM, N, _ = E_hm.shape
        wa, ba, wb, bb, wc, bc = self.gate_params_stacked()
        p, seed, ka, kb = self._gate_dropout((BM, N))
        return MF.gate_scores(E_hm.view(BM * N, -1), wa, ba, wb, bb, wc, bc, p, seed, ka, kb).view(BM, N, self.n_heads)

    def _to_reference_tokens(self, E_hm):
        BM, N, _ = E_hm.shape
        return E_hm.view(BM, N, self.n_heads, -1).permute(0, 1, 3, 2)  # [BM,N,512,H] (strided view)

    def _to_reference_slide(self, pooled_hm):
        lead = pooled_hm.shape[:-1]
        return pooled_hm.view(*lead, self.n_heads, -1).transpose(-1, -2).contiguous()  # [...,512,H]

    def forward_headmajor(self, bags, n_views=1, tok_proj=None, need_tokens=True, stain=None):
        """Fast path used by MADELEINE: returns (pooled_hm [BM,(V,)H*512], E_hm, raw scores [BM,N,H]) and, with tok_proj = (W, bias)
        of a Linear over the head-major token embeddings (MADELEINE's token_projector), its output [BM,N,P] as a fourth result.
        need_tokens = False: the caller does not read E_hm (it is returned as None when the split GEMM mode then keeps E as an image
        only: its LayerNorm kernel writes 4 instead of 8 bytes per element and the pooling kernels read the image)."""
        if self.agg_type != 'regular':
            raise NotImplementedError('Agg type not supported. Options are "regular".')
        act = self.attn[0].activation
        fused = act == 'softmax' and n_views == 1
        tok_on_image = tok_proj is None or MF.split_linear_supported(bags.numel() // bags.shape[-1], tok_proj[0].shape[0],
                                                                     tok_proj[0].shape[1])
        E, e_img = self.embed_tokens_headmajor(bags, return_image=True, want_fp32=need_tokens or not fused or not tok_on_image, stain=stain)
        if self.image_only(E, e_img):
            out = self.pool_headmajor(E, tok_proj=tok_proj, e_img=e_img)
            return (out[0], None, out[1]) + tuple(out[2:])
        if tok_proj is not None:
            if fused:
                pooled, scores, tok = self.pool_headmajor(E, tok_proj=tok_proj, e_img=e_img)   # one autograd node for both consumers of E
                return pooled, E, scores, tok
            with torch.autocast(device_type="cuda", enabled=False):
                tok = MF.linear(E, tok_proj[0], tok_proj[1])
            return self.forward_headmajor_from_tokens(E, n_views, e_img) + (tok,)
        return self.forward_headmajor_from_tokens(E, n_views, e_img)

    def forward_headmajor_from_tokens(self, E, n_views=1, e_img=None):
        act = self.attn[0].activation
        if act == 'softmax' and n_views != 1:
            # intra-modality views (Model.py:419-440): two random halves of the token axis (numpy RNG, as the reference),
            # raw scores re-softmaxed per subset -- pooled by index list inside the fused A2+A3 node
            all_indices = np.arange(E.shape[1])
            np.random.shuffle(all_indices)
            mid = len(all_indices) // 2
            views = tuple(MF.h2d(torch.as_tensor(idx, dtype=torch.int32), E.device) for idx in (all_indices[:mid], all_indices[mid:]))
            pooled, scores = self.pool_headmajor(E, views, e_img=e_img)
            return pooled, E, scores
        if act == 'softmax':
            pooled, scores = self.pool_headmajor(E, e_img=e_img)
        else:
            # non-default activations (abmil.py:56-61): scores from the HIP gate kernel, the elementwise activation in torch,
            # the un-normalised weighted pooling in the HIP pool kernels' linear mode (mdl_abmil_wpool_*)
            scores = self._scores_only(E)
            w = activate(scores.unsqueeze(2), act).squeeze(2)  # elementwise; dim=1 only matters for softmax
            pooled = MF.weighted_pool(E, w)
        if n_views == 1:
            return pooled, E, scores
        # intra-modality views (Model.py:419-440): two random halves, raw scores re-softmaxed per subset
        N = E.shape[1]
        all_indices = np.arange(N)
        np.random.shuffle(all_indices)
        mid = len(all_indices) // 2
        views = [pooled.unsqueeze(1)]
        for idx in (all_indices[:mid], all_indices[mid:]):
            ti = MF.h2d(torch.as_tensor(idx, dtype=torch.long), E.device)
            views.append(MF.softmax_pool(E.index_select(1, ti), scores.index_select(1, ti).contiguous()).unsqueeze(1))
        return torch.cat(views, dim=1), E, scores

    # ------------------------------------------------------------------ reference-shaped API
    def forward(self, bags: torch.Tensor, return_attention: bool = False, return_preattn_feats: bool = False, n_views=1):
        """Model.py:375-451.  slide embeddings [BM,(V,)512,H]; raw attention [BM,N,1,H]; tokens [BM,N,512,H]."""
        pooled, E, scores = self.forward_headmajor(bags, n_views, need_tokens=return_preattn_feats)
        slide = self._to_reference_slide(pooled)
        if return_attention:
            return slide, scores.unsqueeze(2)
        if return_preattn_feats:
            return slide, self._to_reference_tokens(E)
        return slide


class MADELEINE(nn.Module):
    def __init__(self, config, stain_encoding=False):
        super().__init__()
        self.config = config
        self.modalities = config.MODALITIES
        self.stain_encoding = stain_encoding
        # opt-in: encode each distinct all-zero bag of an absent stain once, see _absent_stain_plan
        self.skip_absent_stains = bool(getattr(config, "skip_absent_stains", False))
        if self.stain_encoding:
            self.stain_encoding_dim = 32
            self.embedding = nn.Embedding(len(self.modalities), self.stain_encoding_dim)
        else:
            self.stain_encoding_dim = 0
        if self.config.wsi_encoder == "abmil":
            # (any patch_embedding_dim: widths that are not a multiple of 32 are zero-padded at the encoder's entry, ABMILEmbedder.
            # embed_tokens_headmajor -- round 4 raised here)
            pre_params = {'input_dim': self.config.patch_embedding_dim + self.stain_encoding_dim,
                          'hidden_dim': self.config.wsi_encoder_hidden_dim}
            attention_params = {'model': 'ABMIL',
                                'params': {'input_dim': self.config.wsi_encoder_hidden_dim, 'hidden_dim': 512,
                                           'dropout': True, 'activation': self.config.activation,
                                           'n_heads': self.config.n_heads, 'n_classes': 1}}
            width = attention_params['params']['hidden_dim'] * attention_params['params']['n_heads']
            self.token_projector = nn.Linear(width, 128)
            self.wsi_embedders = ABMILEmbedder(pre_params, attention_params)
            self.projector = nn.Linear(width, attention_params['params']['hidden_dim'])
        else:
            raise ValueError('Unsupported wsi_encoder. Must be "abmil". Now is {}.'.format(self.config.wsi_encoder))

    # ------------------------------------------------------------------ helpers
    def _project_slide(self, pooled_hm):
        """projector Linear(2048, 512) (Model.py:145) on the head-major pooled embeddings (columns permuted to match)."""
        with torch.autocast(device_type="cuda", enabled=False):   # fp32 pooled embeddings: the fp32 kernel in both modes
            return MF.linear(pooled_hm, self.wsi_embedders.permuted(self.projector.weight, 1), self.projector.bias)

    def _project_tokens(self, E_hm):
        """token_projector Linear(2048, 128) (Model.py:140) on the head-major token embeddings."""
        with torch.autocast(device_type="cuda", enabled=False):
            return MF.linear(E_hm, self.wsi_embedders.permuted(self.token_projector.weight, 1), self.token_projector.bias)

    def _stain_groups(self, idx, n_bags, n_tokens, device):
        """(embedding rows [n_bags, 32], row_group int32 [n_bags * n_tokens], cu_groups int64 [n_bags + 1]) for dense bags of n_tokens rows:
        what ABMILEmbedder.embed_tokens_headmajor(stain=...) folds into the first Linear instead of a concatenated copy of the bags."""
        e_rows = self.embedding(MF.h2d(idx, device))
        row_group = torch.arange(n_bags * n_tokens, device=device, dtype=torch.int32).div_(n_tokens, rounding_mode="floor")
        cu = torch.arange(n_bags + 1, device=device, dtype=torch.int64) * n_tokens
        return e_rows, row_group, cu

    @staticmethod
    def _absent_stain_plan(modality_labels, input_key_of_row):
        """the dataset fills a missing stain with an all-zero bag (wsi_dataset.py:66) that the
        reference still pushes through the whole encoder (~27 % of ACROBAT's bags) although the losses never read it.
        All-zero bags with the same stain-encoding index (`input_key_of_row`; one key for all without stain encoding) are
        IDENTICAL inputs, so one representative per key is encoded and its outputs are shared.  Returns (compact, expand): rows of the flattened [B*M] batch to encode, and for
        every original row the position of its source in the compact list; (None, None) when nothing is absent.
        In eval mode every output equals the reference's; in train mode the absent rows share one dropout draw instead of
        having one each -- in outputs that no loss term reads (trainer.py:27-29 gates on the same labels)."""
        present = modality_labels.detach().cpu().reshape(-1).bool()
        if bool(present.all()):
            return None, None
        rows = torch.arange(present.numel())
        compact = rows[present].tolist()
        position = {r: i for i, r in enumerate(compact)}
        representative = {}
        expand = []
        for r in rows.tolist():
            if r in position:
                expand.append(position[r])
                continue
            key = int(input_key_of_row[r])
            if key not in representative:
                representative[key] = len(compact)
                compact.append(r)
            expand.append(representative[key])
        return torch.tensor(compact, dtype=torch.long), torch.tensor(expand, dtype=torch.long)

    @staticmethod
    def _require_single_modality(n_mod):
        # the reference's eval / attention branches reshape with .view(bs*n_mod, d_out*n_heads) on a [bs,512,H]
        # tensor (Model.py:194-196, :210-212): they raise for n_mod != 1.  Same contract here.
        if n_mod != 1:
            raise RuntimeError("MADELEINE.forward(train=False) expects data['feats'] of shape [B, 1, N, D] "
                               "(one modality per call), got n_mod=%d" % n_mod)

    # ------------------------------------------------------------------ public API
    def encode_he(self, feats, device):
        """Model.py:97-107: [B,N,D] -> [B,512]."""
        feats = feats.to(device)
        pooled, _, _ = self.wsi_embedders.forward_headmajor(feats, need_tokens=False)
        return self._project_slide(pooled)

    def encode_he_bags(self, bags, device):
        """encode_he (Model.py:97-107) for SEVERAL bags of different lengths in one launch set: `bags` = list of [N_i, D] (or [1, N_i, D])
        tensors -> [len(bags), 512].  The tokens are packed [sum N_i, D] and pooled through the ragged kernels (cu_seqlens); every kernel
        of the path is row-local and the pooling merges a bag's 128-token chunks in bag order, so each row equals encode_he of that bag
        alone BIT FOR BIT as long as every bag is on the same kernel path alone as in the pack (more than 256 patches: the large-M
        engines; run_inference sends smaller bags one by one).  For the extraction loop (utils.run_inference): one bag per call leaves a
        30,000-patch bag on 235 pooling workgroups and the host bound by ~40 launches per bag."""
        emb = self.wsi_embedders
        if emb.attn[0].activation != 'softmax' or len(bags) == 1:     # (the ragged pooling kernels serve the softmax activation)
            return torch.cat([self.encode_he(b.reshape(1, -1, b.shape[-1]), device) for b in bags])
        flat = [b.reshape(-1, b.shape[-1]) for b in bags]
        lens = [int(x.shape[0]) for x in flat]
        cu = torch.zeros(len(flat) + 1, dtype=torch.int64)
        cu[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int64), 0)
        x = torch.cat([f.to(device) for f in flat], dim=0)
        E, e_img = emb.embed_tokens_headmajor(x, return_image=True, want_fp32=False)
        pooled, _ = emb.pool_headmajor_ragged(E, MF.h2d(cu, device), max(lens), e_img=e_img)
        return self._project_slide(pooled)

    def forward_ragged(self, bags, device, n_loss_tokens=None):
        """Variable-length bags (BASELINE config 5) -- NEW functionality: the reference can only torch.stack equal-N
        bags (wsi_dataset.py:89-92).  `bags` is a list over cases of lists over modalities of [N_bm, D] tensors.
        Semantics = the train branch applied to each bag on its own (incl. the stain-encoding row quirk r // B of
        Model.py:125-131); the pooling kernels take the packed tokens + cu_seqlens, nothing is padded.
        Returns the reference-shaped dicts, with token embeddings restricted to the first `n_loss_tokens` tokens of
        every bag -- all the local loss ever reads (GOT sub-samples randperm(k)[:256])."""
        bs, n_mod = len(bags), len(bags[0])
        if n_loss_tokens is None:
            # GOT draws token indices randperm(k)[:256] with k = the number of participating CASES (reference quirk, loss.py:282,
            # SURVEY.md section 8(a) G0): indices reach k - 1, so a batch of more than 256 cases needs that many tokens kept
            n_loss_tokens = max(256, bs)
        flat = [bags[b][m] for b in range(bs) for m in range(n_mod)]          # case-major rows, like .view(bs*n_mod,...)
        lens = [int(x.shape[0]) for x in flat]
        if min(lens) < n_loss_tokens:
            raise ValueError("every bag needs at least n_loss_tokens=%d tokens (shortest has %d): the local loss reads token "
                             "indices up to min(batch, 256) - 1 of every bag" % (n_loss_tokens, min(lens)))
        cu = torch.zeros(len(flat) + 1, dtype=torch.int64)
        cu[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int64), 0)
        x = torch.cat([f.to(device) for f in flat], dim=0)                     # packed [T, D]
        stain = None
        if self.stain_encoding:
            row_stain = torch.arange(bs * n_mod) // bs                          # the train-branch quirk
            # one embedding row per BAG; the bag of every packed token as an int32 map (folded into the first Linear on the split engine,
            # gathered + concatenated by the other engines: ABMILEmbedder.embed_tokens_headmajor)
            bag_of_tok = MF.h2d(torch.repeat_interleave(torch.arange(bs * n_mod, dtype=torch.int32), torch.tensor(lens)), device)
            stain = (self.embedding(MF.h2d(row_stain, device)), bag_of_tok, None)
        emb = self.wsi_embedders
        cu_d = MF.h2d(cu, device)
        if stain is not None:
            stain = (stain[0], stain[1], cu_d)
        head = MF.h2d((cu[:-1].unsqueeze(1) + torch.arange(n_loss_tokens).unsqueeze(0)).reshape(-1), device)
        tp = (emb.permuted(self.token_projector.weight, 1), self.token_projector.bias)
        # Split GEMM mode: the token projection is part of the pooling node (every token, on the image of E) and the head tokens are
        # gathered from ITS output -- gathering rows of E instead makes autograd fill a zero [T, 2048] tensor and add it to the node's dE
        # (3 x 11 GB of traffic per config-5 step), and E need not exist in fp32 at all.
        fuse_tok = (not bf16_mode()) and MF.split_linear_supported(x.shape[0], tp[0].shape[0], tp[0].shape[1]) \
            and emb.attn[0].activation == 'softmax'      # (the other activations pool fp32 tokens through the weighted-pooling kernels)
        E, e_img = emb.embed_tokens_headmajor(x, return_image=True, want_fp32=not fuse_tok, stain=stain)   # [T, H*512]
        if fuse_tok and e_img is not None:
            pooled, _, tok_all = emb.pool_headmajor_ragged(E, cu_d, max(lens), e_img=e_img, tok_proj=tp)
            tok = tok_all.index_select(0, head).view(bs, n_mod, n_loss_tokens, -1)          # [B,M,n,128]
        else:
            pooled, _ = emb.pool_headmajor_ragged(E, cu_d, max(lens), e_img=e_img)
            tok = self._project_tokens(E.index_select(0, head)).view(bs, n_mod, n_loss_tokens, -1)
        slide = self._project_slide(pooled).view(bs, n_mod, 1, -1)              # [B,M,1,512]
        all_embeddings, all_token_embeddings = {}, {}
        slides, toks = slide.unbind(1), tok.unbind(1)   # (see forward: one stacked gradient instead of per-stain fills and adds)
        for idx, modality in enumerate(self.modalities):
            s, t = slides[idx], toks[idx]
            if modality == "HE":
                s = s.unsqueeze(3).expand(-1, -1, -1, n_mod - 1)
                t = t.unsqueeze(3).expand(-1, -1, -1, n_mod - 1)
            all_embeddings[modality] = s
            all_token_embeddings[modality] = t
        return all_embeddings, all_token_embeddings

    def forward(self, data, device, train=True, n_views=1, custom_stain_idx=None, return_attention=False):
        if 'bags' in data and 'feats' not in data:   # ragged extension (see forward_ragged); keeps DDP's forward hook path
            return self.forward_ragged(data['bags'], device)
        all_wsi_feats = data['feats'].to(device)
        all_embeddings, all_token_embeddings = {}, {}
        emb = self.wsi_embedders

        if train:  # Model.py:120-159
            bs, n_mod, n_tokens, d_in = all_wsi_feats.shape
            x = all_wsi_feats.view(bs * n_mod, n_tokens, d_in)
            # reference quirk kept on purpose (Model.py:125-131): the stain-indicator list is stain-major while the
            # flattened rows are case-major, so row r gets embedding index r // bs.
            stain_of_row = torch.arange(bs * n_mod) // bs
            expand = None
            if self.skip_absent_stains and 'modality_labels' in data:
                compact, expand = self._absent_stain_plan(data['modality_labels'],
                                                          stain_of_row if self.stain_encoding else torch.zeros_like(stain_of_row))
                if expand is not None:   # encode every present bag + ONE all-zero bag per distinct input among the absent ones
                    x = x.index_select(0, MF.h2d(compact, device))
                    stain_of_row = stain_of_row[compact]
            stain = self._stain_groups(stain_of_row, x.shape[0], n_tokens, device) if self.stain_encoding else None
            # token_projector (Model.py:140) inside the pooling node: the two gradients of E are accumulated in the gate dX epilogue
            pooled, _, _, tok = emb.forward_headmajor(x, n_views=n_views, need_tokens=False, stain=stain, tok_proj=(
                emb.permuted(self.token_projector.weight, 1), self.token_projector.bias))   # tok [rows,N,128]
            slide = self._project_slide(pooled.view(x.shape[0], -1, pooled.shape[-1]))  # [rows,V,512]
            if expand is not None:       # absent rows take the outputs of their all-zero representative
                expand = MF.h2d(expand, device)
                tok, slide = tok.index_select(0, expand), slide.index_select(0, expand)
            tok = tok.view(bs, n_mod, n_tokens, -1)                                   # [B,M,N,128]
            slide = slide.view(bs, n_mod, -1, slide.shape[-1])
            # unbind, not M selects: its backward stacks the per-stain gradients in ONE pass; M selects make autograd fill a dense
            # [B,M,N,128] zero tensor per stain and add them pairwise (5 fills + 4 adds of 335 MB each at config 3: 1.1 ms per step)
            slides, toks = slide.unbind(1), tok.unbind(1)
            for idx, modality in enumerate(self.modalities):
                s, t = slides[idx], toks[idx]
                if modality == "HE":
                    # reference: .unsqueeze(3).repeat(1,1,1,M-1); expand gives the same values without copies
                    s = s.unsqueeze(3).expand(-1, -1, -1, n_mod - 1)
                    t = t.unsqueeze(3).expand(-1, -1, -1, n_mod - 1)
                all_embeddings[modality] = s
                all_token_embeddings[modality] = t
            return all_embeddings, all_token_embeddings

        elif not train and not return_attention:  # Model.py:162-203
            bs, n_mod, n_tokens, d_in = all_wsi_feats.shape
            self._require_single_modality(n_mod)
            for stain_idx in range(n_mod):
                stain_name = self.modalities[custom_stain_idx] if custom_stain_idx else self.modalities[stain_idx]
                cur = all_wsi_feats[:, stain_idx]
                stain = None
                if self.stain_encoding:
                    key = custom_stain_idx if custom_stain_idx else stain_idx
                    stain = self._stain_groups(torch.full((bs,), key, dtype=torch.long), bs, n_tokens, device)
                pooled, _, _ = emb.forward_headmajor(cur, need_tokens=False, stain=stain)
                # the reference's .view(bs*n_mod, ...) / .view(bs, n_mod, d) only type-checks for n_mod == 1
                all_embeddings[stain_name] = self._project_slide(pooled).view(bs, n_mod, -1)
            return all_embeddings

        else:  # Model.py:206-216
            bs, n_mod, n_tokens, d_in = all_wsi_feats.shape
            self._require_single_modality(n_mod)
            pooled, _, scores = emb.forward_headmajor(all_wsi_feats[:, HE_POSITION], need_tokens=False)
            he = self._project_slide(pooled).view(bs, n_mod, -1)
            return he, scores.unsqueeze(2)
