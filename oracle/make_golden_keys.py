"""state_dict key/shape lists of the UNMODIFIED reference model classes (build container only).

    python oracle/make_golden_keys.py   ->  tests/golden/ref_state_keys.json

The GPU box has no reference tree, so the on-device model swap (`medaka_amd.integration.convert`,
reference flow datastore.py:135-157 + models.py:392-400) is exercised there on stand-ins
(tests/ref_standins.py).  This file pins the stand-ins to the real classes: same `to_dict()`, same
state_dict keys in the same order, same shapes and dtypes -- checked against the live reference in
tests/test_host.py whenever /root/reference is present.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

CONFIGS = {
    "GRUModel": ("GRUModel", dict(num_features=10, num_classes=5, gru_size=128)),
    "LatentSpaceLSTM": ("LatentSpaceLSTM", dict()),
    "LatentSpaceLSTM_uni": ("LatentSpaceLSTM", dict(bidirectional=False)),
    "rl_lstm384_dwells": ("LatentSpaceLSTM", dict(lstm_size=384, cnn_size=128, use_dwells=True, bidirectional=False)),
    "rl_lstm384_no_dwells": ("LatentSpaceLSTM", dict(lstm_size=384, cnn_size=128, use_dwells=False, bidirectional=False)),
}


def describe(model):
    return {"to_dict": model.to_dict(),
            "state": [[k, list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()]}


def main():
    arch, _, _ = ref_shim.reference_modules()
    out = {name: describe(getattr(arch, cls)(**kw)) for name, (cls, kw) in CONFIGS.items()}
    path = os.path.join(ROOT, "tests", "golden", "ref_state_keys.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(path, {k: len(v["state"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
