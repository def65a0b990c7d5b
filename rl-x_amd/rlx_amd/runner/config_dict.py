"""A small stand-in for `ml_collections.config_dict.ConfigDict` + `config_flags` (the
reference builds its three config namespaces with them: rl_x/runner/runner.py:179-181,
266-270; neither absl nor ml_collections is a dependency of this package).  Supports what
the reference's plugins use: attribute / item access, `in`, `items()`, `to_dict()`,
`str()`, and typed `--<ns>.<key>=<value>` overrides."""


class _ConfigDict:
    def __init__(self, initial=None):
        object.__setattr__(self, "_fields", {})
        for k, v in (initial or {}).items():
            self[k] = v

    def __getattr__(self, name):
        try:
            return object.__getattribute__(self, "_fields")[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self[name] = value

    def __getitem__(self, name):
        return self._fields[name]

    def __setitem__(self, name, value):
        if isinstance(value, dict):
            value = _ConfigDict(value)
        self._fields[name] = value

    def __contains__(self, name):
        return name in self._fields

    def __iter__(self):
        return iter(self._fields)

    def keys(self):
        return self._fields.keys()

    def items(self):
        return list(self._fields.items())

    def get(self, name, default=None):
        return self._fields.get(name, default)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, _ConfigDict) else v) for k, v in self._fields.items()}

    def __str__(self):
        def fmt(d, ind):
            out = []
            for k, v in d._fields.items():
                if isinstance(v, _ConfigDict):
                    out.append(" " * ind + f"{k}:")
                    out.extend(fmt(v, ind + 2))
                else:
                    out.append(" " * ind + f"{k}: {v!r}" if isinstance(v, str) else " " * ind + f"{k}: {v}")
            return out
        return "\n".join(fmt(self, 0))

    __repr__ = __str__


try:  # a genuine rl_x Runner feeds the plugins' default configs to ml_collections' config_flags
    from ml_collections.config_dict import ConfigDict  # noqa: F401
except ImportError:
    ConfigDict = _ConfigDict


def _parse_bool(s):
    if s.lower() in ("true", "1", "yes", "t", "y"):
        return True
    if s.lower() in ("false", "0", "no", "f", "n"):
        return False
    raise ValueError(f"not a boolean: {s!r}")


def cast_like(default, text, flag):
    """Typed override, ml_collections-style: the default's type decides the parse."""
    try:
        if isinstance(default, bool):
            return _parse_bool(text)
        if isinstance(default, int):
            try:
                return int(text)
            except ValueError:
                f = float(text)
                if f != int(f):
                    raise
                return int(f)
        if isinstance(default, float):
            return float(text)
        return text
    except ValueError:
        raise ValueError(f"flag --{flag}={text!r}: expected a value of type {type(default).__name__}") from None


def apply_flag_overrides(namespaces, argv):
    """namespaces: {"runner": ConfigDict, ...}.  Consumes `--ns.key=value` / `--ns.key value`
    / `--ns.bool_key` / `--nons.bool_key`-free forms from argv; raises ValueError on unknown
    flags (absl raises UnrecognizedFlagError).  Returns the set of explicitly set flag names
    (used like `_flagvalues` at rl_x/runner/runner.py:335)."""
    explicit = set()
    i = 1
    while i < len(argv):
        arg = argv[i]
        i += 1
        if not arg.startswith("--"):
            raise ValueError(f"unexpected positional argument {arg!r}")
        body = arg[2:]
        if "=" in body:
            name, text = body.split("=", 1)
        else:
            name, text = body, None
        if "." not in name:
            raise ValueError(f"unknown flag --{name}")
        ns, key = name.split(".", 1)
        if ns not in namespaces or key not in namespaces[ns]:
            raise ValueError(f"unknown flag --{name}")
        default = namespaces[ns][key]
        if text is None:
            if isinstance(default, bool):
                text = "true"
            elif i < len(argv):
                text = argv[i]
                i += 1
            else:
                raise ValueError(f"flag --{name} needs a value")
        namespaces[ns][key] = cast_like(default, text, name)
        explicit.add(name)
    return explicit
