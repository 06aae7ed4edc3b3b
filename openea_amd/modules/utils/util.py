"""Host helpers with the names openea/modules/utils/util.py exports (the TensorFlow session helper of
util.py:7-9 has no counterpart: there is no session, the device state lives in openea_amd.ops)."""
import os
from datetime import datetime


def merge_dic(dic1, dic2):
    """util.py:12-13: the union, entries of `dic2` winning."""
    merged = dict(dic1)
    merged.update(dic2)
    return merged


def task_divide(idx, n):
    """util.py:16-30: `n` consecutive fragments, the first n-1 of len(idx)//n items and the last with the
    remainder; the whole list as one fragment when it cannot be cut into n."""
    count = len(idx)
    if not 0 < n <= count:
        return [idx]
    width = count // n
    cuts = [k * width for k in range(n)] + [count]
    return [idx[lo:hi] for lo, hi in zip(cuts, cuts[1:])]


def generate_out_folder(out_folder, training_data_path, div_path, method_name):
    """util.py:33-39: <out>/<method>/<dataset>/<split><timestamp>/ (same naming, same two log lines)."""
    parts = training_data_path.strip(os.sep).split(os.sep)
    print(out_folder, training_data_path, parts, div_path, method_name)
    stamp = datetime.now().strftime("%Y%m%d%H%M%S")
    folder = "{}{}/{}/{}{}/".format(out_folder, method_name, parts[-1], div_path, stamp)
    print("results output folder:", folder)
    return folder


def load_session():
    """util.py:7-9 returned a tf.Session; the device state lives behind openea_amd.ops, so there is nothing to hand out.
    Kept so that `self.session = load_session()` in code written against the reference keeps working."""
    return None
