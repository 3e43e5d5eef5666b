"""Fixture C1: the PARSED values (yaml.safe_load) of the reference's three config files on the path -- data, not text -- so that
the config boundary ("Config = EasyDict from the YAML ... unchanged") is pinned key by key without the reference
present:  python oracle/gen_golden_configs.py  ->  tests/golden/C1_configs.json"""
import json
import os

import yaml

REF = '/root/reference/tools/cfgs'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'C1_configs.json')
FILES = ('dataset_configs/once_temporal_dataset.yaml', 'once_models/t_mae.yaml', 'once_models/t_mae_ssl.yaml')
json.dump({f: yaml.safe_load(open(os.path.join(REF, f))) for f in FILES}, open(OUT, 'w'), indent=1, sort_keys=True)
print('wrote', OUT)
