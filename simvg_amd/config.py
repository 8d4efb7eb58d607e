"""Python-file configs with the semantics the reference gets from `mmcv.Config` (tools/train.py:190-216,
tools/test.py:117-127): `_base_` inheritance, dict-into-dict merging with `_delete_`, attribute access, dotted
`--cfg-options` overrides and the `key=value` command-line grammar of `mmcv.DictAction`.

mmcv is a third-party dependency of the reference that is not in this image (and not in /root/reference): this file
restates its published behaviour (mmcv 1.x `mmcv/utils/config.py`); the parity tests anchor on how the reference's
own config files and tools use it (`tests/test_apis_cpu.py`) -- parity UNPINNED against mmcv itself.

What is reproduced, because the reference's configs or tools rely on it:
  * a config is a Python file; every top-level name that is not dunder / module / function / class becomes a key
    (so helper variables such as `lr`, `img_size`, `train_pipeline` are keys too, as with mmcv);
  * `_base_ = "x.py"` or a list of files, paths relative to the including file; bases are merged first (a key defined
    by two bases is an error), then the child is merged over them: dict values merge recursively, everything else
    (lists included) is replaced; a child dict containing `_delete_=True` replaces instead of merging;
  * file names may contain characters that are illegal in module names (`noema#finetune#refcoco.py`): the file is
    executed from its text, never imported by name;
  * `{{ fileDirname }}`, `{{ fileBasename }}`, `{{ fileBasenameNoExtension }}`, `{{ fileExtname }}` substitution;
  * `cfg.a.b`, `cfg["a"]["b"]`, `cfg.get`, `cfg.pop` on nested dicts, `cfg.x = v` (dicts become ConfigDicts), missing
    attribute -> AttributeError (so `getattr(cfg.data, "val_flickr30k", None)` and `hasattr(cfg.data, "testA")` work);
  * `merge_from_dict({"a.b.c": v, "data.train.pipeline.0.type": "X"})` (integer components index into lists);
  * `pretty_text` / `dump()` produce a Python file that loads back to the same config.
"""
import argparse
import ast
import copy
import os
import types

BASE_KEY = "_base_"
DELETE_KEY = "_delete_"
RESERVED_KEYS = ("filename", "text", "pretty_text")


class ConfigDict(dict):
    """dict with attribute access; nested dicts are converted on the way in (addict.Dict as mmcv uses it, minus the
    auto-creation of missing keys, which mmcv disables too)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _hook(cls, v):
        if isinstance(v, ConfigDict):
            return v
        if isinstance(v, dict):
            return cls(v)
        if isinstance(v, list):
            return [cls._hook(x) for x in v]
        if isinstance(v, tuple):
            return tuple(cls._hook(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._hook(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'") from None

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name) from None

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, list):
                return [plain(x) for x in v]
            if isinstance(v, tuple):
                return tuple(plain(x) for x in v)
            return v
        return plain(self)


def _merge_a_into_b(a, b, allow_list_keys=False):
    """Merge dict `a` (child / overrides) into a copy of dict `b` (bases).  mmcv `Config._merge_a_into_b`."""
    b = copy.copy(b) if not isinstance(b, list) else list(b)
    for k, v in a.items():
        if allow_list_keys and isinstance(b, list) and k.isdigit():
            idx = int(k)
            if idx >= len(b):
                raise KeyError(f"Index {idx} exceeds the length of list {b}")
            b[idx] = _merge_a_into_b(v, b[idx], allow_list_keys) if isinstance(v, dict) else v
        elif isinstance(v, dict):
            if not isinstance(b, list) and k in b and not v.get(DELETE_KEY, False):
                allowed = (dict, list) if allow_list_keys else dict
                if not isinstance(b[k], allowed):
                    raise TypeError(f"{k}={v} in child config cannot inherit from base because {k} is a dict in the "
                                    f"child config but is of type {type(b[k])} in base config. You may set "
                                    f"`{DELETE_KEY}=True` to ignore the base config.")
                b[k] = _merge_a_into_b(v, b[k], allow_list_keys)
            else:
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
                b[k] = ConfigDict(v)
        else:
            b[k] = v
    return b


def _substitute_predefined(text, filename):
    d = os.path.dirname(filename)
    base = os.path.basename(filename)
    stem, ext = os.path.splitext(base)
    for key, val in (("fileDirname", d), ("fileBasename", base), ("fileBasenameNoExtension", stem), ("fileExtname", ext)):
        for pat in ("{{ " + key + " }}", "{{" + key + "}}"):
            text = text.replace(pat, val.replace("\\", "/"))
    return text


def _file2dict(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(f'file "{filename}" does not exist')
    if not filename.endswith(".py"):
        raise IOError("Only py type are supported now!")
    with open(filename, encoding="utf-8") as f:
        text = f.read()
    src = _substitute_predefined(text, filename)
    ast.parse(src, filename)                      # syntax errors name the real file
    scope = {"__file__": filename, "__name__": "_simvg_cfg_"}
    exec(compile(src, filename, "exec"), scope)
    cfg = {k: v for k, v in scope.items()
           if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    if BASE_KEY in cfg:
        bases = cfg.pop(BASE_KEY)
        bases = bases if isinstance(bases, list) else [bases]
        merged, texts = {}, []
        for b in bases:
            bd, bt = _file2dict(os.path.join(os.path.dirname(filename), b))
            dup = merged.keys() & bd.keys()
            if dup:
                raise KeyError(f"Duplicate key is not allowed among bases. Duplicate keys: {dup}")
            merged.update(bd)
            texts.append(bt)
        cfg = _merge_a_into_b(cfg, merged)
        text = "\n".join(texts + [text])
    return cfg, text


def _fmt(v, indent):
    pad = " " * indent
    if isinstance(v, dict):
        if not v:
            return "dict()"
        if all(isinstance(k, str) and k.isidentifier() for k in v):
            body = ",\n".join(f"{pad}    {k}={_fmt(x, indent + 4)}" for k, x in v.items())
            return "dict(\n" + body + ")"
        body = ",\n".join(f"{pad}    {k!r}: {_fmt(x, indent + 4)}" for k, x in v.items())
        return "{\n" + body + "}"
    if isinstance(v, (list, tuple)):
        o, c = ("[", "]") if isinstance(v, list) else ("(", ",)" if len(v) == 1 else ")")
        if not any(isinstance(x, (dict, list, tuple)) for x in v):
            return o + ", ".join(repr(x) for x in v) + c
        return o + "\n" + ",\n".join(f"{pad}    {_fmt(x, indent + 4)}" for x in v) + "\n" + pad + c
    return repr(v)


class Config:
    """`Config.fromfile(path)`; attribute and item access go to the merged ConfigDict."""

    def __init__(self, cfg_dict=None, cfg_text=None, filename=None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError(f"cfg_dict must be a dict, but got {type(cfg_dict)}")
        for key in cfg_dict:
            if key in RESERVED_KEYS:
                raise KeyError(f"{key} is reserved for config file")
        object.__setattr__(self, "_cfg_dict", ConfigDict(cfg_dict))
        object.__setattr__(self, "_filename", filename)
        object.__setattr__(self, "_text", cfg_text if cfg_text is not None else "")

    @staticmethod
    def fromfile(filename):
        cfg_dict, text = _file2dict(str(filename))
        return Config(cfg_dict, cfg_text=text, filename=str(filename))

    @property
    def filename(self):
        return self._filename

    @property
    def text(self):
        return self._text

    @property
    def pretty_text(self):
        return "\n".join(f"{k} = {_fmt(v, 0)}" for k, v in self._cfg_dict.items()) + "\n"

    def dump(self, file=None):
        if file is None:
            return self.pretty_text
        with open(file, "w", encoding="utf-8") as f:
            f.write(self.pretty_text)

    def merge_from_dict(self, options, allow_list_keys=True):
        option_cfg = {}
        for full_key, v in options.items():
            d = option_cfg
            parts = full_key.split(".")
            for sub in parts[:-1]:
                d = d.setdefault(sub, {})
            d[parts[-1]] = v
        merged = _merge_a_into_b(option_cfg, self._cfg_dict, allow_list_keys=allow_list_keys)
        object.__setattr__(self, "_cfg_dict", ConfigDict(merged))

    # dict / attribute protocol -----------------------------------------------------------
    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def __repr__(self):
        return f"Config (path: {self.filename}): {dict.__repr__(self._cfg_dict)}"

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def pop(self, key, *default):
        return self._cfg_dict.pop(key, *default)

    def keys(self):
        return self._cfg_dict.keys()

    def items(self):
        return self._cfg_dict.items()

    def to_dict(self):
        return self._cfg_dict.to_dict()


class DictAction(argparse.Action):
    """argparse action for `--cfg-options k1=v1 k2=[a,b] k3="[(a,b),(c,d)]"` (mmcv.DictAction grammar): ints, floats,
    true/false, None, strings; comma lists, [..] lists, (..) tuples, nested."""

    @staticmethod
    def _parse_int_float_bool(val):
        for cast in (int, float):
            try:
                return cast(val)
            except ValueError:
                pass
        if val.lower() in ("true", "false"):
            return val.lower() == "true"
        if val == "None":
            return None
        return val

    @staticmethod
    def _parse_iterable(val):
        def find_next_comma(string):
            assert string.count("(") == string.count(")") and string.count("[") == string.count("]"), \
                f"Imbalanced brackets exist in {string}"
            end = len(string)
            for idx, char in enumerate(string):
                pre = string[:idx]
                if char == "," and pre.count("(") == pre.count(")") and pre.count("[") == pre.count("]"):
                    end = idx
                    break
            return end

        val = val.strip("'\"").replace(" ", "")
        is_tuple = False
        if val.startswith("(") and val.endswith(")"):
            is_tuple = True
            val = val[1:-1]
        elif val.startswith("[") and val.endswith("]"):
            val = val[1:-1]
        elif "," not in val:
            return DictAction._parse_int_float_bool(val)
        values = []
        while len(val) > 0:
            comma = find_next_comma(val)
            values.append(DictAction._parse_iterable(val[:comma]))
            val = val[comma + 1:]
        return tuple(values) if is_tuple else values

    def __call__(self, parser, namespace, values, option_string=None):
        options = {}
        for kv in values:
            key, val = kv.split("=", maxsplit=1)
            options[key] = self._parse_iterable(val)
        setattr(namespace, self.dest, options)
