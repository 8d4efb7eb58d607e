"""tests/golden/config_models.json: the `model` dict of every experiment config of the reference (configs/**/*.py, 53 files)
as the reference's own config files evaluate (read with simvg_amd.config.Config, whose mmcv-Config semantics are pinned in
tests/test_apis_cpu.py), keyed by the path relative to configs/.  DATA, no reference text: what `build_model(cfg.model)`
receives for each experiment -- the repo's models must be constructible from all of them (tests/test_configs_cpu.py),
which the GPU box (no /root/reference) can then check too.

    python -m oracle.make_golden_configs          (dev container only: needs /root/reference)"""
import glob
import json
import os

REF = "/root/reference/configs"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config_models.json")


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def main():
    from simvg_amd.config import Config
    out = {}
    for f in sorted(glob.glob(os.path.join(REF, "**", "*.py"), recursive=True)):
        if "/_base_/" in f:
            continue
        cfg = Config.fromfile(f)
        out[os.path.relpath(f, REF)] = dict(model=_plain(cfg.model.to_dict() if hasattr(cfg.model, "to_dict") else dict(cfg.model)),
                                            ema=bool(cfg.get("ema", False)), dataset=str(cfg.get("dataset", "")))
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(f"wrote {OUT}: {len(out)} configs")


if __name__ == "__main__":
    main()
