"""Convolution solver selection for the trunk ("rest PyTorch-ROCm": MIOpen).

Out of the box PyTorch asks MIOpen for its *heuristic* solver per convolution.  For the NHWC bf16 convolutions of this graph at
640 px / 16 images that choice leans on split-K implicit-GEMM kernels wrapped in zero-fill and cast kernels (589 launches, 5 ms per
step) and costs 10 ms per step against what MIOpen's own timed search picks (105.8 -> 95.6 ms per step).  The search takes ~4
minutes on a fresh machine, so its result - MIOpen's text find-db / perf-db, 130 KB - is shipped under tuned/miopen/ and handed to
MIOpen through MIOPEN_USER_DB_PATH (a writable copy in a persistent per-user, per-MIOpen-build directory: MIOpen locks and appends
to it).  Shapes that are not in the table (an epoch's tail batch, which the loader keeps like the reference's does; the validation batch)
are searched once per machine and kept there.
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


def _user_db_dir(suffix=''):
    """A PERSISTENT per-user directory for MIOpen's writable tables, one per MIOpen / HIP build: what a run has to search (a tail
    batch, the validation batch size, another image size) is searched once per machine, not once per run, and nothing is left behind in
    /tmp.  TAMTR_MIOPEN_DB_DIR overrides the place; an unwritable home falls back to a temporary directory removed at exit."""
    tag = f'miopen-{torch.backends.cudnn.version()}-hip-{torch.version.hip}{suffix}'
    base = os.environ.get('TAMTR_MIOPEN_DB_DIR') or os.path.join(os.environ.get('XDG_CACHE_HOME') or os.path.join(os.path.expanduser('~'), '.cache'),
                                                                 'tamtr_amd')
    path = os.path.join(base, tag)
    try:
        os.makedirs(path, exist_ok=True)
        probe = os.path.join(path, f'.w{os.getpid()}')
        open(probe, 'w').close()
        os.remove(probe)
        return path, True
    except OSError:
        import atexit
        tmp = tempfile.mkdtemp(prefix='tamtr_miopen_')
        atexit.register(shutil.rmtree, tmp, ignore_errors=True)
        return tmp, False


def _seed(dst):
    """Copy the shipped tables into dst unless it already holds them (MIOpen appends its own finds to these files: never overwrite a
    file that has grown).  Ranks of one job race here harmlessly: the copy goes through a temporary name + rename."""
    for f in glob.glob(os.path.join(_DIR, '*.txt')):
        to = os.path.join(dst, os.path.basename(f))
        if os.path.exists(to) and os.path.getsize(to) >= os.path.getsize(f):
            continue
        tmp = f'{to}.{os.getpid()}.tmp'
        shutil.copy(f, tmp)
        os.replace(tmp, to)


def use_deterministic_convolutions(search=False):
    """The reference's `deterministic: True` (cfg/default.yaml:26 -> utils/torch_utils.py:371-389: cudnn.deterministic +
    use_deterministic_algorithms(warn_only)) for the MIOpen part of the step: ATen sets MIOpen's DETERMINISTIC convolution attribute,
    under which the split-K / atomic-add solvers are not applicable (and rocBLAS atomics are off for torch's own GEMMs).  For NHWC bf16 convolutions that leaves MIOpen 3.5 with its naive
    kernels only (17 s per 16-image step), so the deterministic mode runs the trunk NCHW (model._CHANNELS_LAST / set_channels_last(False):
    0.25 s per step eager).  search=False (the default, and what TAMTR_DETERMINISTIC=1 uses): MIOpen's heuristic among the solvers that
    remain.  search=True: its timed search among them, kept in a persistent directory of its own (the shipped tables were chosen with the
    atomic solvers allowed and are not used) - 20 % faster (0.20 s per step with graph replay) but NOT reliably reproducible, see
    use_tuned_convolutions.  Call before the first convolution."""
    os.environ['MIOPEN_DEBUG_CONVOLUTION_DETERMINISTIC'] = '1'
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = bool(search)
    torch.use_deterministic_algorithms(True, warn_only=True)
    if search:
        work, _ = _user_db_dir('-deterministic')
        os.environ['MIOPEN_USER_DB_PATH'] = work
        return f'deterministic (MIOpen timed search among its deterministic solvers, {work})'
    return 'deterministic (MIOpen heuristic restricted to deterministic solvers)'


def _no_naive_solvers():
    """MIOpen's "naive" reference convolutions (ConvDirectNaiveConv{Fwd,Bwd,Wrw}: one thread per output element, double accumulation) are
    applicable to every problem, so a timed search TIMES them for every convolution it looks at - 0.3 ... 690 ms per call, 66 s of GPU
    time before the first step of a 640 px / 16 image run (profiles/r03_bench_kernel_stats.csv rows 1-4: 368 calls each of the fwd /
    bwd / wrw kernels), find-db hit or not.  They never win, so the search modes take them out of the candidate list (MIOpen reads the
    switches on its first use of them: call before the first convolution; exported values win).  Measured (profiles/r04_startup.txt):
    fp32 probe forward 29.9 -> 2.2 s, capture warm-up 47.4 -> 0.9 s, step time unchanged.  NOT done outside the search modes: MIOpen's
    heuristic picks the naive kernels for fp32 NHWC maps, and the fp32 parity tests are pinned on their double-accumulated sums (with
    the next-best fp32 solvers one class logit of 17 520 moves by 1.4e-3 relative, past the 1e-3 bound of
    test_full_model_640_fp32_elementwise_with_the_oracles_choices); and the deterministic mode needs them (the only NHWC bf16 solvers
    that qualify)."""
    for d in ('FWD', 'BWD', 'WRW'):
        os.environ.setdefault(f'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_{d}', '0')


def use_tuned_convolutions(mode='shipped', db_dir=None, log=None, rank_suffix=''):
    """Call before the first convolution.  mode: 'shipped' - MIOpen's immediate mode on the shipped tables (the best recorded solution per
    convolution, nothing timed in this process) if they match this MIOpen build, else MIOpen's default heuristic; 'search' - timed search into db_dir (slow first run; how the tables are made);
    'off' - default heuristic.  TAMTR_DETERMINISTIC=1 overrides all of them with use_deterministic_convolutions().
    Returns what was set up, for logs."""
    if os.environ.get('TAMTR_DETERMINISTIC') == '1':
        # (no timed search: measured on MI355X, a search under the DETERMINISTIC attribute came back with a solver set that was bitwise
        # reproducible in one run and not in the next (whole-gradient difference between two eager steps 0 vs 1.5e-2) - some solver it times
        # passes the attribute and still sums in a run-dependent order; MIOpen's heuristic under the attribute has been reproducible in every run)
        return use_deterministic_convolutions(search=False)
    if mode == 'off':
        return 'off (MIOpen heuristic)'
    if mode == 'search':
        db_dir = db_dir or _user_db_dir(rank_suffix)[0]
        os.makedirs(db_dir, exist_ok=True)
        os.environ['MIOPEN_USER_DB_PATH'] = db_dir
        torch.backends.cudnn.benchmark = True
        _no_naive_solvers()
        return f'search ({db_dir})'
    if mode != 'shipped':
        raise ValueError(mode)
    if not shipped_db_matches():
        return 'off (shipped tables are for another MIOpen build)'
    work, persistent = _user_db_dir(rank_suffix)
    _seed(work)
    os.environ['MIOPEN_USER_DB_PATH'] = work
    # MIOpen's IMMEDIATE mode on the tables (cudnn.benchmark off): every convolution takes the best RECORDED solution; nothing is timed in this
    # process.  With cudnn.benchmark on, ATen calls MIOpen's Find, which re-times the candidates in every process whatever the tables hold (it
    # rewrote the user copy of the find-db in every run): the solver choice then depends on that run's timings - identical on an idle GPU, but
    # with several ranks timing on one GPU at once some weight-gradient convolutions came back on memset-based solvers (round 4's 2-rank
    # rehearsal, profiles/r04_ddp_memset_probe.txt), a different set on every rank and run.  Same step time (77.5 against 77.6 ms on one box).
    # TAMTR_CONV_FIND=search restores the timed search (what 'search' mode uses to write tables for other shapes).
    torch.backends.cudnn.benchmark = os.environ.get('TAMTR_CONV_FIND') == 'search'
    _no_naive_solvers()
    if log:
        log(f'convolutions: MIOpen immediate mode on the shipped tables (640 px / 16 images, 1280 px / 8 images) in {work}'
            f'{"" if persistent else " (temporary)"}; a shape outside them - a tail batch, the validation batch - gets MIOpen\'s heuristic choice '
            '(tools/tune_miopen.sh / --conv-tuning search writes tables for other shapes)')
    return 'shipped tables'


def use_tuned_convolutions_ranked(mode='shipped', db_dir=None, log=None, rank=0, world=1):
    """use_tuned_convolutions() for one rank of a data-parallel job on one host.  Every rank gets a table directory OF ITS OWN
    (`...-rank<r>`, seeded from the shipped tables): MIOpen guards its writable tables with lock files, and a rank that does not get the
    lock in time treats the lookup as a miss - it then runs a real timed search for that convolution and may come back with another
    solver than the table holds (round 4's 2-rank rehearsal: seven weight-gradient convolutions on memset-based solvers on one rank, which
    the graph recorder refuses under AQL packet capture - profiles/r04_ddp_memset_probe.txt).  world == 1 is the plain call (shared
    per-user directory)."""
    if world <= 1:
        return use_tuned_convolutions(mode, db_dir, log)
    sub = None if db_dir is None else os.path.join(db_dir, f'rank{rank}')
    return use_tuned_convolutions(mode, sub, log, rank_suffix=f'-rank{rank}')
