"""Convolution solver selection for the trunk ("rest PyTorch-ROCm": MIOpen).

Out of the box PyTorch asks MIOpen for its *heuristic* solver per convolution.  For the NHWC bf16 convolutions of this graph at
640 px / 16 images that choice leans on split-K implicit-GEMM kernels wrapped in zero-fill and cast kernels (589 launches, 5 ms per
step) and costs 10 ms per step against what MIOpen's own timed search picks (105.8 -> 95.6 ms per step).  The search takes ~4
minutes on a fresh machine, so its result - MIOpen's text find-db / perf-db, 130 KB - is shipped under tuned/miopen/ and handed to
MIOpen through MIOPEN_USER_DB_PATH (a writable copy: MIOpen locks and appends to it).  Shapes that are not in the table are searched
once and added to the copy.
"""
import glob
import json
import os
import shutil
import tempfile

import torch

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned', 'miopen')


def shipped_db_matches():
    """The shipped tables were written by this MIOpen / HIP version (the file names MIOpen looks for carry its version)."""
    try:
        meta = json.load(open(os.path.join(_DIR, 'meta.json')))
    except OSError:
        return False
    return (bool(glob.glob(os.path.join(_DIR, '*.ufdb.txt'))) and meta.get('miopen_version') == torch.backends.cudnn.version()
            and meta.get('hip') == torch.version.hip)


def use_tuned_convolutions(mode='shipped', db_dir=None):
    """Call before the first convolution.  mode: 'shipped' - timed-search mode backed by the shipped tables if they match this
    MIOpen build, else MIOpen's default heuristic; 'search' - timed search into db_dir (slow first run; how the tables are made);
    'off' - default heuristic.  Returns what was set up, for logs."""
    if mode == 'off':
        return 'off (MIOpen heuristic)'
    if mode == 'search':
        db_dir = db_dir or tempfile.mkdtemp(prefix='tamtr_miopen_')
        os.makedirs(db_dir, exist_ok=True)
        os.environ['MIOPEN_USER_DB_PATH'] = db_dir
        torch.backends.cudnn.benchmark = True
        return f'search ({db_dir})'
    if mode != 'shipped':
        raise ValueError(mode)
    if not shipped_db_matches():
        return 'off (shipped tables are for another MIOpen build)'
    work = tempfile.mkdtemp(prefix='tamtr_miopen_')
    for f in glob.glob(os.path.join(_DIR, '*.txt')):
        shutil.copy(f, work)
    os.environ['MIOPEN_USER_DB_PATH'] = work
    torch.backends.cudnn.benchmark = True
    return 'shipped tables'
