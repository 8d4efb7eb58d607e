"""Model registries and build_* helpers -- the plugin API of the hot path.

Mirror of the reference's `simvg/models/builder.py:4-36` (mmcv `Registry` x5 + `build_model(cfg,
word_emb=None, num_token=-1)`), re-implemented without mmcv: `Registry.build(cfg, default_args)` pops
`type`, fills defaults, instantiates; classes self-register with `@X.register_module()`.
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def __contains__(self, key):
        return key in self.module_dict

    def __repr__(self):
        return f"Registry(name={self.name}, items={sorted(self.module_dict)})"

    def get(self, key):
        return self.module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _do(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self.module_dict[key] = cls
            return cls

        return _do(module) if module is not None else _do

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise KeyError(f"`cfg` for registry {self.name} must be a dict with the key 'type', got {cfg!r}")
        kwargs = dict(cfg)
        for k, v in (default_args or {}).items():
            kwargs.setdefault(k, v)
        typ = kwargs.pop("type")
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return cls(**kwargs)


class skip_init:
    """`with skip_init(): model = build_model(cfg)` -- the large random initialisations (truncated-normal encoder weights, Xavier
    head weights) are left as uninitialised memory.  ONLY for callers that load every parameter right afterwards
    (`load_state_dict(strict=True)`): building ViT-L otherwise spends tens of seconds of single-threaded CPU time on
    values that are overwritten."""
    active = False

    def __enter__(self):
        self._prev, skip_init.active = skip_init.active, True
        return self

    def __exit__(self, *exc):
        skip_init.active = self._prev


VIS_ENCODERS = Registry("VIS_ENCS")
LAN_ENCODERS = Registry("LAN_ENCS")
MODELS = Registry("MODELS")
FUSIONS = Registry("FUSIONS")
HEADS = Registry("HEADS")


def build_vis_enc(cfg):
    return VIS_ENCODERS.build(cfg)


def build_lan_enc(cfg, default_args):
    return LAN_ENCODERS.build(cfg, default_args=default_args)


def build_fusion(cfg):
    return FUSIONS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_model(cfg, word_emb=None, num_token=-1):
    return MODELS.build(cfg, default_args=dict(word_emb=word_emb, num_token=num_token))
