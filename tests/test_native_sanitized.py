"""The product's own GPU-free host code under AddressSanitizer + UBSan and ThreadSanitizer (CPU only: sanitizers are not
available on the GPU pool): the pool of copy threads (kpal_amd/csrc/host_pool.hpp), the host side of the FASTA ingest --
source, read-ahead, chunk cutting, the state carried across chunk seams (kpal_amd/csrc/fasta_host.hpp) -- and the copy phase
of the CPython gatherer behind Profile.from_sequences (kpal_amd/csrc/kpal_join_core.h), and the stream / event schedule of the
pipelined multi-GPU table reduce (kpal_amd/csrc/comm_schedule.hpp) driven by a fake runtime.  The harnesses are tests/native/*."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, 'tests', 'native')

SAN = {
    'asan': ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined'],
    'tsan': ['-fsanitize=thread'],
}

CASES = [
    # (harness, compiler, language flags, pool sizes to run with)
    ('host_pool_check.cpp', 'g++', ['-std=c++17'], ['1', '2', '16']),
    ('fasta_host_check.cpp', 'g++', ['-std=c++17'], ['1', '5']),
    ('join_check.c', 'gcc', ['-std=c11'], ['']),
    ('gather_check.c', 'gcc', ['-std=c11'], ['']),      # kpal_amd/csrc/kpal_gather_core.h: walk AND copies on several threads
    # the stream / event schedule of the pipelined multi-GPU reduce on a fake runtime: ranks, streams and events as threads
    ('comm_schedule_check.cpp', 'g++', ['-std=c++17'], ['']),
]


@pytest.mark.parametrize('san', sorted(SAN))
@pytest.mark.parametrize('case', CASES, ids=[c[0].split('.')[0] for c in CASES])
def test_host_code_sanitized(case, san, tmp_path):
    src, cc, lang, pools = case
    if shutil.which(cc) is None:
        pytest.skip('no %s' % cc)
    exe = str(tmp_path / (src.split('.')[0] + '_' + san))
    build = subprocess.run([cc, '-O1', '-g', '-fno-omit-frame-pointer'] + lang + SAN[san] + ['-o', exe, os.path.join(NATIVE, src), '-lpthread'],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if build.returncode != 0 and b'sanitize' in build.stdout.lower() and (b'cannot find' in build.stdout or b'not supported' in build.stdout):
        pytest.skip('sanitizer runtime not installed: %s' % build.stdout.decode()[-300:])
    assert build.returncode == 0, build.stdout.decode()[-2000:]
    for pool in pools:
        env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1', TSAN_OPTIONS='halt_on_error=1 exitcode=66', KPAL_READ_PIN='0')
        env.pop('LD_PRELOAD', None)
        if pool:
            env['KPAL_READ_THREADS'] = pool
        run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
        out = run.stdout.decode()
        if run.returncode != 0 and 'unexpected memory mapping' in out:
            pytest.skip('ThreadSanitizer cannot map its shadow here (ASLR setting of the container)')
        assert run.returncode == 0 and 'SANITIZE_OK' in out, 'pool %r: %s' % (pool, out[-3000:])
