"""BatchedABMIL -- gated attention head, mirror of the reference class of the same name
(reference madeleine/models/abmil.py:8-68): same constructor, same sub-module names (hence the same
state_dict keys attention_a.0.*, attention_b.0.*, attention_c.*), same forward contract.

The arithmetic runs in libmadeleine_amd.so: the fused gate kernels (mdl_abmil_gate_fwd/bwd and their split-engine siblings) are
specialised for the geometry the MADELEINE encoder hard-wires (input_dim = hidden_dim = 512, n_classes = 1, Model.py:64-77);
any other geometry -- including the class defaults 1024 / 256 / 1 -- runs its two Linears on the HIP Linear kernels with elementwise
activations (`_forward_generic`).  Inside ABMILEmbedder only the fused geometry is accepted.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import functional as MF

GATE_DROPOUT_P = 0.25  # abmil.py:33-35


class BatchedABMIL(nn.Module):
    def __init__(self, input_dim=1024, hidden_dim=256, dropout=False, n_classes=1, n_heads=1, activation='softmax'):
        super().__init__()
        self.activation = activation
        self.input_dim, self.hidden_dim, self.n_classes = input_dim, hidden_dim, n_classes
        self.use_dropout = bool(dropout)
        a = [nn.Linear(input_dim, hidden_dim), nn.Tanh()]
        b = [nn.Linear(input_dim, hidden_dim), nn.Sigmoid()]
        if dropout:
            a.append(nn.Dropout(GATE_DROPOUT_P))
            b.append(nn.Dropout(GATE_DROPOUT_P))
        self.attention_a = nn.Sequential(*a)
        self.attention_b = nn.Sequential(*b)
        self.attention_c = nn.Linear(hidden_dim, n_classes)
        self._injected_keep = None  # (keep_a, keep_b) uint8 [B,N,512]: explicit dropout masks (parity tests)

    # -- parameters in the layout the C ABI takes ([H,512,512] etc. with H = 1 for a lone head)
    def gate_params(self):
        return (self.attention_a[0].weight, self.attention_a[0].bias, self.attention_b[0].weight,
                self.attention_b[0].bias, self.attention_c.weight.view(-1), self.attention_c.bias.view(-1))

    def _check_geometry(self):
        if self.input_dim != MF.HID or self.hidden_dim != MF.HID or self.n_classes != 1:
            raise NotImplementedError(
                "madeleine_amd.BatchedABMIL: the HIP gate kernels are specialised for input_dim = hidden_dim = 512, "
                "n_classes = 1 (the geometry MADELEINE hard-wires, Model.py:64-77); got (%d, %d, %d)"
                % (self.input_dim, self.hidden_dim, self.n_classes))

    def dropout_p(self) -> float:
        """Rate of the two nn.Dropout modules of the gate (abmil.py:33-35: 0.25 each) while training, else 0."""
        if not (self.use_dropout and self.training):
            return 0.0
        pa, pb = float(self.attention_a[2].p), float(self.attention_b[2].p)
        if pa != pb:
            raise NotImplementedError("madeleine_amd.BatchedABMIL: the gate kernels take one dropout rate for both branches")
        return pa

    def _is_fused_geometry(self):
        return self.input_dim == MF.HID and self.hidden_dim == MF.HID and self.n_classes == 1

    def _forward_generic(self, x, return_raw_attention):
        """Any other geometry -- e.g. the class defaults input_dim = 1024, hidden_dim = 256 (abmil.py:10): the two gate Linears run on
        the HIP Linear kernels (functional.linear: split-fp16 / fp32 matrix-core engine, forward + dX + dW), the tanh / sigmoid / dropout
        and the hidden -> n_classes product (a reduction over <= hidden_dim channels per class) are elementwise torch ops."""
        B, N, D = x.shape
        if D != self.input_dim:
            raise ValueError("BatchedABMIL: expected input dim %d, got %d" % (self.input_dim, D))
        x2 = x.float().contiguous().view(B * N, D)
        a = torch.tanh(MF.linear(x2, self.attention_a[0].weight, self.attention_a[0].bias))
        b = torch.sigmoid(MF.linear(x2, self.attention_b[0].weight, self.attention_b[0].bias))
        p = self.dropout_p()
        if p > 0:
            if self._injected_keep is not None:
                ka, kb = (k.reshape(B * N, self.hidden_dim).to(a.dtype) for k in self._injected_keep)
                a, b = a * ka / (1.0 - p), b * kb / (1.0 - p)
            else:
                a, b = F.dropout(a, p, True), F.dropout(b, p, True)
        ab = a * b
        A = torch.stack([(ab * self.attention_c.weight[c]).sum(-1) + self.attention_c.bias[c] for c in range(self.n_classes)], dim=-1)
        A = A.view(B, N, self.n_classes)
        activated = activate(A, self.activation)
        return (activated, A) if return_raw_attention else activated

    def forward(self, x, return_raw_attention=False):
        """x [B, N, input_dim] -> activated attention [B, N, n_classes] (and the raw scores when asked)."""
        if x.dim() != 3:
            raise ValueError("BatchedABMIL expects x of shape [batch, tokens, dim]")
        if not self._is_fused_geometry():
            return self._forward_generic(x, return_raw_attention)
        B, N, D = x.shape
        wa, ba, wb, bb, wc, bc = self.gate_params()
        p = self.dropout_p()
        ka = kb = None
        if p > 0 and self._injected_keep is not None:
            ka, kb = (k.reshape(B * N, 1, MF.HID).contiguous() for k in self._injected_keep)
        seed = MF.new_dropout_seed() if (p > 0 and ka is None) else 0
        A = MF.gate_scores(x.float().contiguous().view(B * N, D), wa.unsqueeze(0), ba.unsqueeze(0), wb.unsqueeze(0),
                           bb.unsqueeze(0), wc.unsqueeze(0), bc, p, seed, ka, kb).view(B, N, 1)
        activated = activate(A, self.activation)
        if return_raw_attention:
            return activated, A
        return activated


def activate(A: torch.Tensor, activation: str) -> torch.Tensor:
    """abmil.py:54-63 -- softmax is over the patch axis (dim=1)."""
    if activation == 'softmax':
        return F.softmax(A, dim=1)
    if activation == 'leaky_relu':
        return F.leaky_relu(A)
    if activation == 'relu':
        return F.relu(A)
    if activation == 'sigmoid':
        return torch.sigmoid(A)
    raise NotImplementedError('Activation not implemented.')
