"""Minimal gin-config subset (gin is not installed on the target image).

Covers exactly what the reference's co3d_3d/configs/*.gin and train.py:256-257 use:
``Name.param = <python literal>`` bindings (comments, multi-line lists/tuples, last binding
wins), ``@gin.configurable`` on functions and classes, ``parse_config_files_and_bindings``,
``query_parameter("Name.param")`` (reference modules/optim.py:106-111,
classification_training.py:13) and ``clear_config``.  No macros / references / scopes /
includes: none appear in the reference's config files.
"""
import ast
import functools
import inspect

_BINDINGS = {}  # "Name" -> {param: value}
_REGISTRY = {}  # "Name" -> callable


class GinError(ValueError):
    pass


def clear_config():
    _BINDINGS.clear()


def bind_parameter(binding_key, value):
    name, _, param = binding_key.rpartition(".")
    if not name or not param:
        raise GinError(f"malformed binding key {binding_key!r}")
    _BINDINGS.setdefault(name.split("/")[-1], {})[param] = value


def query_parameter(binding_key):
    name, _, param = binding_key.rpartition(".")
    try:
        return _BINDINGS[name][param]
    except KeyError:
        raise GinError(f"Configurable {name!r} has no bound parameter {param!r}") from None


def _statements(text):
    """Join physical lines into logical statements (open brackets continue a statement)."""
    buf, depth = "", 0
    for raw in text.splitlines():
        line = _strip_comment(raw).rstrip()
        if not line.strip() and depth == 0:
            continue
        buf += line + "\n"
        depth += sum(line.count(c) for c in "([{") - sum(line.count(c) for c in ")]}")
        if depth <= 0:
            yield buf.strip()
            buf, depth = "", 0
    if buf.strip():
        raise GinError(f"unterminated statement: {buf!r}")


def _strip_comment(line):
    quote = None
    for i, ch in enumerate(line):
        if quote:
            if ch == quote and line[i - 1] != "\\":
                quote = None
        elif ch in "'\"":
            quote = ch
        elif ch == "#":
            return line[:i]
    return line


def parse_config(text):
    if isinstance(text, (list, tuple)):
        text = "\n".join(text)
    for st in _statements(text):
        key, eq, val = st.partition("=")
        if not eq:
            raise GinError(f"cannot parse gin statement {st!r} (only `Name.param = literal` is supported)")
        try:
            value = ast.literal_eval(val.strip())
        except (ValueError, SyntaxError) as e:
            raise GinError(f"value of {key.strip()!r} is not a python literal: {val.strip()!r}") from e
        bind_parameter(key.strip(), value)


def parse_config_file(path):
    with open(path) as f:
        parse_config(f.read())


def parse_config_files_and_bindings(config_files, bindings, finalize_config=True, skip_unknown=False):
    for f in config_files or []:
        parse_config_file(f)
    parse_config(list(bindings or []))


def _inject(name, fn, args, kwargs):
    bound = _BINDINGS.get(name)
    if not bound:
        return kwargs
    sig = inspect.signature(fn)
    params = sig.parameters
    accepts_kw = any(p.kind == p.VAR_KEYWORD for p in params.values())
    try:
        given = set(sig.bind_partial(*args, **kwargs).arguments)
    except TypeError:
        given = set(kwargs)
    out = dict(kwargs)
    for k, v in bound.items():
        if k in given:
            continue
        if k not in params and not accepts_kw:
            raise GinError(f"configurable {name!r} has no parameter {k!r}")
        out[k] = v
    return out


def configurable(obj=None, name=None, **_ignored):
    """Decorator: parameters not passed by the caller are filled from the bindings."""

    def wrap(target):
        reg = name or target.__name__
        if inspect.isclass(target):
            orig_init = target.__init__

            @functools.wraps(orig_init)
            def __init__(self, *args, **kwargs):
                orig_init(self, *args, **_inject(reg, functools.partial(orig_init, self), args, kwargs))

            target.__init__ = __init__
            _REGISTRY[reg] = target
            return target

        @functools.wraps(target)
        def wrapper(*args, **kwargs):
            return target(*args, **_inject(reg, target, args, kwargs))

        _REGISTRY[reg] = wrapper
        return wrapper

    if callable(obj):
        return wrap(obj)
    if isinstance(obj, str):
        name = obj
    return wrap


def operative_config_str():
    lines = []
    for n in sorted(_BINDINGS):
        for p, v in sorted(_BINDINGS[n].items()):
            lines.append(f"{n}.{p} = {v!r}")
    return "\n".join(lines)
