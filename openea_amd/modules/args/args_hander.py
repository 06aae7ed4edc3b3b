"""JSON -> attribute object: the reference's flag system (openea/modules/args/args_hander.py)."""
import json


class ARGs:
    """args_hander.py:13-16: one attribute per JSON key."""

    def __init__(self, dic):
        for k, v in dic.items():
            setattr(self, k, v)

    def __repr__(self):
        return "ARGs(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.__dict__.items()))


def load_args(file_path):
    """args_hander.py:4-10."""
    with open(file_path, 'r') as f:
        args_dict = json.load(f)
    print("load arguments:", args_dict)
    return ARGs(args_dict)


def check_args(args):
    """args_hander.py:19-21."""
    assert args.sampling_mode in ["uniform", "truncated"]
    return True
