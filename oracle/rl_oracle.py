"""CPU oracle of the read-level model (test infrastructure; never imported by medaka_amd/).

Restates `LatentSpaceLSTM.forward` (reference medaka/architectures/latent_space_lstm.py:154-207)
with its helpers `ReadLevelConv` / `make_1dconv_layers` (read_level_modules.py:7-78), `MeanPooler`
(read_level_modules.py:81-100) and `ReversibleLSTM` (latent_space_lstm.py:11-33) as plain PyTorch-CPU
functional calls on a `state_dict`, so that it can run on the GPU box where the reference tree is
absent.  Pinned against the unmodified reference in tests/test_oracle_rl.py (goldens made by
oracle/make_golden_rl.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _lstm_layer(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of torch.nn.LSTM (batch_first), gate order i, f, g, o; h0 = c0 = 0."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    out = torch.empty(B, T, H)
    gi_all = x @ w_ih.T + b_ih
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = gi_all[:, t] + (h @ w_hh.T + b_hh)
        i, f, gg, o = g.split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, t] = h
    return out


def rl_forward(x, state, use_dwells=False, bidirectional=True, normalise=True, eps=1e-5):
    """x: uint8 (B, P, D, F) read-level features -> (B, P, 5) float32 probabilities."""
    st = {k: torch.as_tensor(np.asarray(v)) for k, v in state.items()}
    x = torch.as_tensor(np.asarray(x))
    with torch.inference_mode():
        mask = x.sum((1, -1)) != 0                                    # (B, D)
        emb = st["base_embedder.weight"][x[..., 0].long()] + \
            st["strand_embedder.weight"][x[..., 2].long() + 1]
        q = (x[..., 1] / 25 - 1).unsqueeze(-1)
        feats = [emb, q]
        if use_dwells:
            feats.append(x[..., 4].unsqueeze(-1).to(torch.float32))
        y = torch.cat(feats, dim=-1)                                   # (B, P, D, 7|8)
        y = y.permute(0, 2, 3, 1)                                      # (B, D, f, P)
        b, d, f, p = y.shape
        y = y.flatten(0, 1)
        for conv_i, bn_i in ((0, 2), (3, 5)):
            w = st[f"read_level_conv.convs.{conv_i}.weight"]
            y = F.conv1d(y, w, st[f"read_level_conv.convs.{conv_i}.bias"], padding=(w.shape[-1] - 1) // 2)
            y = torch.relu(y)
            y = F.batch_norm(y, st[f"read_level_conv.convs.{bn_i}.running_mean"],
                             st[f"read_level_conv.convs.{bn_i}.running_var"],
                             st[f"read_level_conv.convs.{bn_i}.weight"],
                             st[f"read_level_conv.convs.{bn_i}.bias"], training=False, eps=eps)
        y = y.permute(0, 2, 1)                                         # (B*D, P, C)
        y = y @ st["pre_pool_expansion_layer.weight"].T + st["pre_pool_expansion_layer.bias"]
        y = y.view(b, d, p, -1)
        depth = mask.sum(-1)
        y = (y * mask[..., None, None]).sum(dim=1) / depth[..., None, None]      # (B, P, H)
        if bidirectional:
            for layer in range(2):
                outs = []
                for sfx, rev in (("", False), ("_reverse", True)):
                    outs.append(_lstm_layer(y, st[f"lstm.weight_ih_l{layer}{sfx}"], st[f"lstm.weight_hh_l{layer}{sfx}"],
                                            st[f"lstm.bias_ih_l{layer}{sfx}"], st[f"lstm.bias_hh_l{layer}{sfx}"], rev))
                y = torch.cat(outs, -1)
        else:
            for i in range(4):   # reverse-forward-reverse-forward (latent_space_lstm.py:141-149)
                y = _lstm_layer(y, st[f"lstm.{i}.lstm.weight_ih_l0"], st[f"lstm.{i}.lstm.weight_hh_l0"],
                                st[f"lstm.{i}.lstm.bias_ih_l0"], st[f"lstm.{i}.lstm.bias_hh_l0"], not bool(i % 2))
        y = y @ st["linear.weight"].T + st["linear.bias"]
        if normalise:
            y = torch.softmax(y, dim=-1)
    return y.numpy()


# synthetic inputs / weights live with the product's other generators; re-exported for the tests
from medaka_amd.synth import synth_reads, synth_rl_state  # noqa: E402,F401
