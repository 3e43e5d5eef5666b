"""OpenPCDet config API: global ``cfg``, ``cfg_from_yaml_file`` (with ``_BASE_CONFIG_`` include),
``cfg_from_list`` (``--set A.B v`` overrides), ``log_config_to_file``.  Same behaviour as the reference's
pcdet/config.py:7-107, written against a local attribute-dict (easydict is not a dependency here)."""
from ast import literal_eval
from pathlib import Path

import yaml


class EasyDict(dict):
    """dict with attribute access, recursively applied to nested dicts / lists of dicts."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {}, **kwargs)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    __setattr__ = __setitem__

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, other=None, **kwargs):
        for k, v in dict(other or {}, **kwargs).items():
            self[k] = v


def log_config_to_file(cfg, pre='cfg', logger=None):
    for key, val in cfg.items():
        if isinstance(val, EasyDict):
            logger.info('\n%s.%s = edict()' % (pre, key))
            log_config_to_file(val, pre=pre + '.' + key, logger=logger)
            continue
        logger.info('%s.%s: %s' % (pre, key, val))


def cfg_from_list(cfg_list, config):
    """Set config keys via list (e.g. from the command line: --set MODEL.VFE.TYPE mean)."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        keys = k.split('.')
        d = config
        for sub in keys[:-1]:
            assert sub in d, 'NotFoundKey: %s' % sub
            d = d[sub]
        sub = keys[-1]
        assert sub in d, 'NotFoundKey: %s' % sub
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        if type(value) != type(d[sub]) and isinstance(d[sub], EasyDict):
            for src in value.split(','):
                ck, cv = src.split(':')
                d[sub][ck] = type(d[sub][ck])(cv)
        elif type(value) != type(d[sub]) and isinstance(d[sub], list):
            vals = value.split(',')
            d[sub] = [type(d[sub][0])(x) for x in vals]
        else:
            assert type(value) == type(d[sub]), \
                'type {} does not match original type {}'.format(type(value), type(d[sub]))
            d[sub] = value


def _resolve_base(path, search_dirs):
    """The reference resolves _BASE_CONFIG_ against the cwd (tools/); also try the including file's tree."""
    if Path(path).exists():
        return path
    for d in search_dirs:
        d = Path(d)
        for _ in range(4):
            if (d / path).exists():
                return str(d / path)
            d = d.parent
    return path


def merge_new_config(config, new_config, search_dirs=()):
    if '_BASE_CONFIG_' in new_config:
        with open(_resolve_base(new_config['_BASE_CONFIG_'], search_dirs), 'r') as f:
            config.update(EasyDict(yaml.safe_load(f)))
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config:
            config[key] = EasyDict()
        merge_new_config(config[key], val, search_dirs)
    return config


def cfg_from_yaml_file(cfg_file, config):
    with open(cfg_file, 'r') as f:
        new_config = yaml.safe_load(f)
    merge_new_config(config=config, new_config=new_config, search_dirs=(Path(cfg_file).resolve().parent,))
    return config


cfg = EasyDict()
cfg.ROOT_DIR = (Path(__file__).resolve().parent / '../').resolve()
cfg.LOCAL_RANK = 0
