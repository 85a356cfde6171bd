"""CPU emulation of the ENGINE's half-precision scheme for the rl_lstm384 architecture (not the reference's): is the
deviation the GPU shows for the no-dwells flavour (up to 2.4 x the mean of the reference's own fp16 emulation) inherent
to the scheme -- fp16 operands of every matrix product, fp32 accumulate, fp32 gates and cell state, h rounded to fp16 only
where it feeds a product -- or a defect of the kernels?

    python profiles/r3_experiments/emulate_rl_half.py  ->  emulate_rl_half.txt
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rl_oracle  # noqa: E402


def r16(t):
    return t.to(torch.float16).to(torch.float32)


def engine_half(x, state, use_dwells, eps=1e-5):
    st = {k: torch.as_tensor(np.asarray(v)) for k, v in state.items()}
    x = torch.as_tensor(np.asarray(x))
    with torch.inference_mode():
        mask = x.sum((1, -1)) != 0
        emb = st["base_embedder.weight"][x[..., 0].long()] + st["strand_embedder.weight"][x[..., 2].long() + 1]
        feats = [emb, (x[..., 1] / 25 - 1).unsqueeze(-1)]
        if use_dwells:
            feats.append(x[..., 4].unsqueeze(-1).float())
        y = torch.cat(feats, -1).permute(0, 2, 3, 1)
        b, d, f, p = y.shape
        y = y.flatten(0, 1)
        bn = lambda y, i: F.batch_norm(y, st[f"read_level_conv.convs.{i}.running_mean"], st[f"read_level_conv.convs.{i}.running_var"],
                                       st[f"read_level_conv.convs.{i}.weight"], st[f"read_level_conv.convs.{i}.bias"], training=False, eps=eps)
        y = bn(torch.relu(F.conv1d(y, st["read_level_conv.convs.0.weight"], st["read_level_conv.convs.0.bias"])), 2)   # conv1: fp32
        w2 = st["read_level_conv.convs.3.weight"]
        y = F.conv1d(r16(y).double(), r16(w2).double(), None, padding=8).float() + st["read_level_conv.convs.3.bias"][None, :, None]
        y = bn(torch.relu(y), 5).permute(0, 2, 1).reshape(b, d, p, -1)
        depth = mask.sum(-1)
        pooled = (y * mask[..., None, None]).sum(1) / depth[..., None, None]                     # (B, P, 128) fp32
        we, be = st["pre_pool_expansion_layer.weight"].double(), st["pre_pool_expansion_layer.bias"].double()
        h_in = pooled
        for i in range(4):
            w_ih, w_hh = st[f"lstm.{i}.lstm.weight_ih_l0"], st[f"lstm.{i}.lstm.weight_hh_l0"]
            bias = (st[f"lstm.{i}.lstm.bias_ih_l0"] + st[f"lstm.{i}.lstm.bias_hh_l0"]).double()
            if i == 0:                                  # Linear(128 -> 384) folded into W_ih in double precision
                wf = (w_ih.double() @ we).float()
                bias = bias + w_ih.double() @ be
            else:
                wf = w_ih
            gi = (r16(h_in).double() @ r16(wf).double().T + bias).float()
            B, T, H = gi.shape[0], gi.shape[1], w_hh.shape[1]
            h = torch.zeros(B, H)
            c = torch.zeros(B, H)
            out = torch.empty(B, T, H)
            whh = r16(w_hh).double().T
            for t in (range(T - 1, -1, -1) if i % 2 == 0 else range(T)):      # reverse-forward-reverse-forward
                g = gi[:, t] + (r16(h).double() @ whh).float()
                ig, fg, gg, og = g.split(H, 1)
                c = torch.sigmoid(fg) * c + torch.sigmoid(ig) * torch.tanh(gg)
                h = torch.sigmoid(og) * torch.tanh(c)
                out[:, t] = h
            h_in = out
        y = h_in @ st["linear.weight"].T + st["linear.bias"]
        return torch.softmax(y, -1).numpy()


if __name__ == "__main__":
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    lines = []
    for flavour, dw, seed in (("dwells", True, 21), ("no dwells", False, 22)):
        kw = dict(lstm_size=384, cnn_size=128, use_dwells=dw, bidirectional=False)
        st = rl_oracle.synth_rl_state(seed=seed, **kw)
        for B, P, D in ((5, 300, 6), (40, 130, 3), (300, 20, 2)):
            x = rl_oracle.synth_reads(B, P, D, use_dwells=dw, seed=B + P)
            ref = rl_oracle.rl_forward(x, st, use_dwells=dw, bidirectional=False)
            emu = engine_half(x, st, dw)
            dd = np.abs(emu - ref)
            lines.append(f"{flavour:10s} {B}x{P}x{D}: engine-scheme emulation vs fp32: max|dp| {dd.max():.2e}  mean {dd.mean():.2e}  "
                         f"argmax agreement {(emu.argmax(-1) == ref.argmax(-1)).mean():.4f}")
            print(lines[-1], flush=True)
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "emulate_rl_half.txt"), "w").write("\n".join(lines) + "\n")
