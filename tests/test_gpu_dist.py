"""The multi-GPU plumbing on one GPU: a zero-copy torch view of the device count table and RCCL
collectives on it (world size 1 -- the 8-GPU run is the driver's).  Run on the GPU box."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_collectives_on_table_view():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', GRAFT_REPO_ROOT=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'nccl_view_check.py')], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-2000:]
    assert 'NCCL_VIEW_OK' in out


def test_bench_single_gpu_small():
    """bench.py contract on a small workload: one JSON line with roofline and a true checksum."""
    import json
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1',
                        '--reads', '2000000', '--cpu-reads', '200000', '--cpu-big-reads', '1000000', '--profile-reads', '100000'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line
    assert line['checksum_ok'] and line['n_gpus'] == 1 and line['unit'] == 'Gbases/s'
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(line['roofline'])
    assert line['cpu_baseline']['kind'] == 'port' and line['cpu_baseline']['cores'] == 1
    assert line['rccl_ranks'] == 1 and line['roofline']['pipeline']['frac'] > 0
    # the extras of the default run: BASELINE configs 4 and 5 and the end-to-end figure, each with its own parity flag
    ex = line['extra']
    for name in ('end_to_end', 'k15', 'matrix_prod', 'matrix_euclidean'):
        assert 'error' not in ex[name], (name, ex[name])
        assert ex[name]['checksum_ok'] is True, (name, ex[name])
    assert ex['k15']['ms_per_step'] > 0 and ex['k15']['roofline']['pipeline']['frac'] > 0
    assert ex['matrix_prod']['parity_pairs'] == 28 and ex['matrix_euclidean']['roofline']['bound'] == 'mfma'
    assert ex['end_to_end']['h2d_s'] > 0 and ex['end_to_end']['d2h_s'] > 0
    # skew is a first-class figure of the line: uniform / 2 % low-complexity / homopolymer input at k = 12 and k = 15, as scalars of `config`
    for name in ('k12_uniform', 'k12_low_complexity_2pct', 'k12_homopolymer', 'k15_uniform', 'k15_low_complexity_2pct', 'k15_homopolymer'):
        assert ex['skew'][name]['Gbases_per_s'] > 0 and line['config']['skew_%s_Gbases_per_s' % name] == ex['skew'][name]['Gbases_per_s']


def test_library_rccl_world_1():
    """The in-library multi-GPU entry points on one GPU (world size 1; the 8-GPU run is the driver's): communicator from a
    unique id, kpal_comm_reduce_table in its serial and its pipelined form (second stream, side buffers alternating over
    three steps) + balance, against oracle.balance(oracle.count).  In a child process: a communicator that cannot be
    built must not take the test session down."""
    env = dict(os.environ, GRAFT_REPO_ROOT=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rccl_library_check.py')], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-3000:]
    assert 'RCCL_LIBRARY_OK' in out


def test_bench_multi_gpu_code_path_on_one_gpu():
    """The N > 1 code of bench.py -- rank supervisor, worker process, the library's RCCL communicator and torch.distributed's,
    the three reduce modes (pipelined / serial in-library reduce, torch reduce) each timed with its own warm-up and steps, the
    per-bin comparison of the merged + balanced table with the single-stream count, the line's fields -- started exactly as
    the driver starts the scaling runs (python -m torch.distributed.run ... bench.py --gpus N ...), with a world of ONE rank
    (KPAL_BENCH_FORCE_MULTI=1: this box has one GPU; the 8-GPU run is the driver's)."""
    import json
    env = dict(os.environ, KPAL_BENCH_FORCE_MULTI='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                        '--master-port', '29633', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--reads', '3000000'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line['n_gpus'] == 1 and line['rccl_ranks'] == 1 and line['attempt'] == 1
    assert line['reduce_mode'] == 'library_pipelined' and line['pipelined_reduce'] is True and line['reduce_via'] == 'library'
    assert line['merged_equals_single_stream'] is True and line['checksum_ok'] is True
    assert line['config']['merged_equals_single_stream'] is True and line['config']['reduce_mode'] == 'library_pipelined'
    for mode in ('pipelined_reduce', 'serial_reduce', 'torch_reduce', 'range_merge'):
        assert line['extra'][mode]['merged_equals_single_stream'] is True and line['extra'][mode]['ms_per_step'] > 0, mode
        assert line['config'][mode + '_ms_per_step'] > 0
    # (at 3 M reads the fixed-cost histogram stage may be the longest kernel: an intermediate kernel has no algorithmic bytes of its own)
    assert line['value'] > 0 and line['roofline']['pipeline_frac'] > 0 and 'rccl_reduce' in line['roofline']['kernels_ms_per_step']
    # the N > 1 line carries the CPU baseline too (rank 0's host cores, the N = 1 sample)
    assert 'cpu_baseline' in line and line['cpu_baseline']['value'] > 0 and line['cpu_baseline']['kind'] == 'port'
    assert isinstance(line['config'].get('rccl_info'), str)       # (what RCCL said it chose: rank 0's NCCL_DEBUG=INFO log, condensed)


def test_bench_two_ranks_sharing_one_gpu():
    """TWO ranks with real GPU workers, started as the driver starts the scaling runs.  This box has one GPU and RCCL refuses two ranks on
    one device, so the ranks share device 0 and talk over gloo (KPAL_BENCH_SHARED_GPU=1): what runs is everything of the N > 1 path
    but the RCCL collectives themselves -- two supervisors, two workers, the shards of a world of two (rank 1 counts reads
    [R, 2R)), torch.distributed's reduce of the two tables to rank 0 + balance, the max-over-ranks timing, and the per-bin comparison
    of the merged table with the single-stream count of both shards."""
    import json
    env = dict(os.environ, KPAL_BENCH_SHARED_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29641', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--reads', '1500000',
                        '--no-cpu'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['attempt'] == 1 and line['config']['shared_gpu'] is True
    assert line['reduce_mode'] == 'torch_serial' and line['reduce_via'] == 'torch'
    assert line['merged_equals_single_stream'] is True and line['checksum_ok'] is True
    assert line['config']['reads_per_gpu'] == 1500000 and line['value'] > 0
    assert 'KPAL_BENCH_SHARED_GPU' in line['library_rccl_error']


def _build_fake_rccl(tmp_path):
    lib = str(tmp_path / 'libfake_rccl.so')
    subprocess.run([os.environ.get('HIPCC', 'hipcc'), '-O2', '-shared', '-fPIC', '-o', lib, os.path.join(ROOT, 'tests', 'native', 'fake_rccl.cpp')],
                   check=True, timeout=600)
    return lib


def _start_world(lib, id_file, world, **extra_env):
    env = dict(os.environ, KPAL_RCCL_LIBRARY=lib, KPAL_FAKE_RCCL_TIMEOUT_S='240', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.update(extra_env)
    return [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'rccl_world_rank.py'), str(r), str(world), id_file], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]


def _finish_world(procs):
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0].decode(errors='replace'))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return procs, outs


_ASYNC = {'KPAL_FAKE_RCCL_ASYNC': '1', 'KPAL_FAKE_RCCL_DELAY_MS': '15'}


@pytest.fixture(scope='module')
def worlds(tmp_path_factory):
    """The five worlds of rccl_world_rank.py the tests below look at -- three with a fault switch of the stand-in, the healthy ones
    of 2 and 4 ranks -- started TOGETHER (sixteen processes on the one GPU; each world has its own shared-memory file): run one
    after the other they spent a minute of the GPU suite's time box mostly asleep in the stand-in's bounded waits."""
    tmp = tmp_path_factory.mktemp('worlds')
    lib = _build_fake_rccl(tmp)
    started = {}
    for fault in ('reduce', 'recv', 'early'):
        extra = dict(_ASYNC) if fault == 'early' else {}
        started['fault-' + fault] = _start_world(lib, str(tmp / ('id_' + fault)), 2, KPAL_FAKE_RCCL_FAULT=fault, KPAL_FAKE_RCCL_TIMEOUT_S='20', **extra)
    for world in (2, 4):
        started['async-%d' % world] = _start_world(lib, str(tmp / ('id_w%d' % world)), world, **_ASYNC)
    return {name: _finish_world(procs) for name, procs in started.items()}


@pytest.mark.parametrize('fault', ['reduce', 'recv', 'early'])
def test_fake_rccl_world_test_has_teeth(worlds, fault):
    """The test below must notice a transport that loses a rank's contribution to a reduce, delivers the wrong block of the
    mirrored-range exchange, or (asynchronous mode) lets the stream go on before the collective's data has arrived -- what a missing
    dependency between streams amounts to: with the stand-in told to do so, some rank has to fail an assertion (and none may hang)."""
    procs, outs = worlds['fault-' + fault]
    assert any(p.returncode != 0 for p in procs), outs
    assert any('AssertionError' in out for out in outs), outs


@pytest.mark.parametrize('mode', ['async'])       # ('sync' -- every collective completed inside the call -- passes too and is the schedule with fewer ways to go wrong: tools/multi_rank_one_gpu.sh runs both)
@pytest.mark.parametrize('world', [2, 4])
def test_library_comm_world_over_fake_rccl(worlds, world, mode):
    """The library's kpal_comm_* protocol between W real processes on the one GPU of this box: RCCL refuses two ranks on one device,
    so KPAL_RCCL_LIBRARY points at a stand-in for the dozen entry points the library binds (tests/native/fake_rccl.cpp: shared
    memory between the processes, bounded waits).  Contexts, streams, events, kernels, offsets, the order of the collectives:
    all real (tests/rccl_world_rank.py says what is compared with the oracle).  async: the stand-in's calls return at once, as
    RCCL's do -- the stream is held by a host function until the communicator's worker thread has moved the data, (rank + 1) x 15 ms
    late, so every collective is in flight while the caller goes on queueing work (the pipelined reduce: the next count)."""
    procs, outs = worlds['async-%d' % world]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and 'RCCL_WORLD_OK rank %d of %d' % (r, world) in out, 'rank %d:\n%s' % (r, out[-3000:])


@pytest.mark.parametrize('mode', ['async'])       # (as above)
def test_bench_two_ranks_library_modes_over_fake_rccl(tmp_path, mode):
    """bench.py's N > 1 path with TWO real ranks AND the library's reduce modes: as test_bench_two_ranks_sharing_one_gpu, with the
    stand-in of tests/native/fake_rccl.cpp behind the library's communicator -- pipelined and serial in-library reduce, the torch
    reduce and the bin-range merge are each measured and each merged table is compared bin for bin with the single-stream count."""
    import json
    env = dict(os.environ, KPAL_BENCH_SHARED_GPU='1', KPAL_RCCL_LIBRARY=_build_fake_rccl(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY='0')
    if mode == 'async':
        env.update(KPAL_FAKE_RCCL_ASYNC='1', KPAL_FAKE_RCCL_DELAY_MS='10')
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29647' if mode == 'sync' else '29649', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--reads', '1500000',
                        '--no-cpu'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['attempt'] == 1 and line['config']['shared_gpu'] is True
    assert line['reduce_mode'] == 'library_pipelined' and line['reduce_via'] == 'library' and 'library_rccl_error' not in line
    assert line['merged_equals_single_stream'] is True and line['checksum_ok'] is True
    for mode in ('pipelined_reduce', 'serial_reduce', 'torch_reduce', 'range_merge'):
        assert line['extra'][mode]['merged_equals_single_stream'] is True and line['extra'][mode]['checksum_ok'] is True, mode


@pytest.mark.parametrize('mode', ['async'])      # ('sync' passes too: the host thread hangs inside the call instead of the stream; 30 s more)
def test_bench_supervisors_restart_after_a_hung_collective(tmp_path, mode):
    """A collective of the library's communicator that never completes, with REAL GPU workers: the last rank of two stops inside its
    fourth operation (KPAL_FAKE_RCCL_FAULT=stall; async: its stream stands still behind the operation, sync: its host thread does),
    the other waits for it.  The supervisors must notice (KPAL_BENCH_RUN_TIMEOUT), end both workers -- one of them with work queued
    on the GPU that will never run --, start fresh ones in the conservative mode (torch reduce, no library communicator) and rank 0
    must print ONE verified line that says so."""
    import json
    env = dict(os.environ, KPAL_BENCH_SHARED_GPU='1', KPAL_RCCL_LIBRARY=_build_fake_rccl(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY='0',
               KPAL_FAKE_RCCL_FAULT='stall', KPAL_FAKE_RCCL_TIMEOUT_S='600', KPAL_BENCH_RUN_TIMEOUT='15')
    if mode == 'async':
        env.update(KPAL_FAKE_RCCL_ASYNC='1')
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29651' if mode == 'sync' else '29653', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--reads', '1500000', '--no-cpu'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line['attempt'] == 2 and line['reduce_mode'] == 'torch_serial' and line['reduce_via'] == 'torch'
    assert 'did not finish within' in line['fallback_reason'] or 'gave up' in line['fallback_reason'], line['fallback_reason']
    assert line['merged_equals_single_stream'] is True and line['checksum_ok'] is True and line['n_gpus'] == 2


def test_bench_comm_init_that_never_returns(tmp_path):
    """ncclCommInitRank of the library's communicator hangs on every rank (the last rank never joins: KPAL_FAKE_RCCL_FAULT=init).  It is
    time-boxed on its own (KPAL_BENCH_COMM_INIT_TIMEOUT): the workers give up with the reason, the supervisors start the
    conservative attempt, rank 0 prints one verified line."""
    import json
    env = dict(os.environ, KPAL_BENCH_SHARED_GPU='1', KPAL_RCCL_LIBRARY=_build_fake_rccl(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY='0',
               KPAL_FAKE_RCCL_FAULT='init', KPAL_FAKE_RCCL_TIMEOUT_S='600', KPAL_BENCH_COMM_INIT_TIMEOUT='8')
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29659', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--reads', '1500000', '--no-cpu'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line['attempt'] == 2 and line['reduce_mode'] == 'torch_serial'
    assert 'kpal_comm_init' in line['fallback_reason'] or 'gave up' in line['fallback_reason'], line['fallback_reason']
    assert line['merged_equals_single_stream'] is True and line['checksum_ok'] is True
