"""Host helpers (mirror of openea/modules/utils/util.py, without the TensorFlow import)."""
import time


def merge_dic(dic1, dic2):
    """util.py:12-13."""
    return {**dic1, **dic2}


def task_divide(idx, n):
    """util.py:16-30: n-1 fragments of len(idx)//n items, the last takes the remainder."""
    total = len(idx)
    if n <= 0 or total == 0 or n > total:
        return [idx]
    if n == total:
        return [[i] for i in idx]
    j = total // n
    tasks = [idx[i:i + j] for i in range(0, (n - 1) * j, j)]
    tasks.append(idx[(n - 1) * j:])
    return tasks


def generate_out_folder(out_folder, training_data_path, div_path, method_name):
    """util.py:33-39 (same folder naming, same log lines)."""
    params = training_data_path.strip('/').split('/')
    print(out_folder, training_data_path, params, div_path, method_name)
    path = params[-1]
    folder = out_folder + method_name + '/' + path + "/" + div_path + str(time.strftime("%Y%m%d%H%M%S")) + "/"
    print("results output folder:", folder)
    return folder
