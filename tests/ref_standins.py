"""Stand-ins for the reference model classes on boxes without the reference tree (the GPU box).

`medaka_amd.integration.convert` receives whatever `ModelStoreTGZ.load_model` built (reference
datastore.py:135-157, models.py:392-400): an object whose CLASS NAME, `to_dict()`, attributes and
`state_dict()` it reads.  These classes reproduce exactly that surface with stock torch modules -- they are
parameter containers, their forward is never called -- and tests/test_host.py pins them to the real classes
(tests/golden/ref_state_keys.json from oracle/make_golden_keys.py, plus a live comparison when
/root/reference is present).  Test infrastructure only.
"""
import torch


class _Base(torch.nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self._kwargs = dict(kwargs)
        self.half_precision = False
        self.normalise = True

    def device(self):
        return next(self.parameters()).device

    def half(self):
        super().half()
        self.half_precision = True
        return self

    def to_dict(self):
        return {"type": type(self).__name__, "kwargs": dict(self._kwargs)}

    def forward(self, x):
        raise AssertionError("stand-in: the PyTorch forward must never run")

    predict_on_batch = forward


class GRUModel(_Base):
    def __init__(self, num_features=10, num_classes=5, gru_size=128, n_layers=2, bidirectional=True,
                 time_steps=None, classify_activation=None):
        super().__init__(num_features=num_features, num_classes=num_classes, gru_size=gru_size, n_layers=n_layers,
                         bidirectional=bidirectional, time_steps=time_steps, classify_activation=classify_activation)
        self.num_features, self.num_classes, self.gru_size = num_features, num_classes, gru_size
        self.n_layers, self.bidirectional = n_layers, bidirectional
        self.gru = torch.nn.GRU(num_features, gru_size, num_layers=n_layers, bidirectional=bidirectional, batch_first=True)
        self.linear = torch.nn.Linear((2 if bidirectional else 1) * gru_size, 5)


class _Rev(torch.nn.Module):
    def __init__(self, size):
        super().__init__()
        self.lstm = torch.nn.LSTM(size, size, batch_first=True)


class _Conv(torch.nn.Module):
    def __init__(self, nf, out_dim, kernel_sizes, ch):
        super().__init__()
        mods, cin = [], nf
        for k in kernel_sizes:
            mods += [torch.nn.Conv1d(cin, ch, k, padding=(k - 1) // 2), torch.nn.ReLU(), torch.nn.BatchNorm1d(ch)]
            cin = ch
        self.convs = torch.nn.Sequential(*mods)
        self.expansion_layer = torch.nn.Linear(ch, out_dim)


class LatentSpaceLSTM(_Base):
    def __init__(self, num_classes=5, lstm_size=128, cnn_size=128, kernel_sizes=[1, 17], pooler_type="mean",
                 pooler_args={}, use_dwells=False, bases_alphabet_size=6, bases_embedding_size=6,
                 bidirectional=True, time_steps=None):
        super().__init__(num_classes=num_classes, lstm_size=lstm_size, cnn_size=cnn_size, kernel_sizes=kernel_sizes,
                         pooler_type=pooler_type, pooler_args=pooler_args, use_dwells=use_dwells,
                         bases_alphabet_size=bases_alphabet_size, bases_embedding_size=bases_embedding_size,
                         bidirectional=bidirectional, time_steps=time_steps)
        for k, v in self._kwargs.items():
            if k != "time_steps":
                setattr(self, k, v)
        self.base_embedder = torch.nn.Embedding(bases_alphabet_size, bases_embedding_size)
        self.strand_embedder = torch.nn.Embedding(3, bases_embedding_size)
        self.read_level_conv = _Conv(bases_embedding_size + (2 if use_dwells else 1), lstm_size, kernel_sizes, cnn_size)
        self.pre_pool_expansion_layer = torch.nn.Linear(cnn_size, lstm_size)
        if bidirectional:
            self.lstm = torch.nn.LSTM(lstm_size, lstm_size, num_layers=2, bidirectional=True, batch_first=True)
        else:
            self.lstm = torch.nn.Sequential(*[_Rev(lstm_size) for _ in range(4)])
        self.linear = torch.nn.Linear((2 if bidirectional else 1) * lstm_size, num_classes)


# the configurations of tests/golden/ref_state_keys.json
CONFIGS = {
    "GRUModel": (GRUModel, dict(num_features=10, num_classes=5, gru_size=128)),
    "LatentSpaceLSTM": (LatentSpaceLSTM, dict()),
    "LatentSpaceLSTM_uni": (LatentSpaceLSTM, dict(bidirectional=False)),
    "rl_lstm384_dwells": (LatentSpaceLSTM, dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False)),
    "rl_lstm384_no_dwells": (LatentSpaceLSTM, dict(lstm_size=384, cnn_size=128, use_dwells=False, bidirectional=False)),
}


def describe(model):
    return {"to_dict": model.to_dict(),
            "state": [[k, list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()]}
