"""Minimal stand-in for the `omegaconf` package (not installed in this image).

TEST INFRASTRUCTURE ONLY.  It exists so that the unmodified reference backbone
(/root/reference/models/detection/recurrent_backbone/maxvit_rnn.py:5 imports
``DictConfig, OmegaConf``) can be imported in the authoring container to pin the
oracle.  Only the handful of behaviours the reference backbone touches are
provided: attribute access, ``.get``, ``OmegaConf.create/to_container/is_config``
and ``open_dict``.
"""
from contextlib import contextmanager


class DictConfig(dict):
    def __init__(self, content=None, **kw):
        super().__init__()
        content = dict(content or {}, **kw)
        for k, v in content.items():
            self[k] = _wrap(v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = _wrap(value)


class ListConfig(list):
    pass


def _wrap(v):
    if isinstance(v, DictConfig):
        return v
    if isinstance(v, dict):
        return DictConfig(v)
    if isinstance(v, (list, tuple)) and not isinstance(v, ListConfig):
        return ListConfig(_wrap(x) for x in v)
    return v


def _unwrap(v):
    if isinstance(v, dict):
        return {k: _unwrap(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_unwrap(x) for x in v]
    return v


class OmegaConf:
    @staticmethod
    def create(content=None):
        return _wrap(content if content is not None else {})

    @staticmethod
    def to_container(cfg, resolve=True, throw_on_missing=False):
        return _unwrap(cfg)

    @staticmethod
    def is_config(obj):
        return isinstance(obj, (DictConfig, ListConfig))


@contextmanager
def open_dict(cfg):
    yield cfg
