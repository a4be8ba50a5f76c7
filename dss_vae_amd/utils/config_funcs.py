"""yaml <-> Namespace (reference: dss_vae/utils/config_funcs.py:6-40; safe_load instead of the bare
yaml.load that fails on PyYAML >= 6)."""
from argparse import Namespace

import yaml


def yaml_to_dict(yaml_in_file):
    with open(yaml_in_file) as f:
        return yaml.safe_load(f.read())


def dict_to_yaml(yaml_out_file, dicts):
    with open(yaml_out_file, "w") as f:
        yaml.safe_dump(dicts, f, default_flow_style=False)


def dict_to_args(d):
    return Namespace(**d)


def args_to_dict(args):
    return dict(vars(args))


def args_to_yaml(yaml_out_file, args):
    dict_to_yaml(yaml_out_file=yaml_out_file, dicts=args_to_dict(args))


def yaml_to_args(yaml_in_file):
    return dict_to_args(yaml_to_dict(yaml_in_file=yaml_in_file))
