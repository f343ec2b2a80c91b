"""CPU checks of the C-ABI boundary: the library loads and exports every symbol
include/algp_hip.h declares; no compute call is made (no GPU here)."""
import os
import re

import pytest

from algp_amd import _hip

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(REPO, 'include', 'algp_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(algp_[a-z0-9_]+)\s*\(', txt)))


def test_header_and_binding_agree():
    decl = _declared_symbols()
    assert len(decl) >= 35
    assert sorted(_hip.SIGNATURES) == decl


def test_library_exports_every_declared_symbol():
    lib = _hip.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.algp_version() >= 100


def test_no_silent_fallback_without_gpu():
    if _hip.device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(_hip.AlgpError):
        _hip.Context()


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, 'algp_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(root, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_missing_rccl_is_an_error_return_not_a_crash():
    """$ALGP_RCCL_PATH names a file that does not exist (and, being set, is the only file tried): the communicator entry
    points return an error code -- the process stays alive, the library stays usable (dlerror() was called
    twice and a null char* went into std::string)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from algp_amd import _hip\n"
            "lib = _hip.load()\n"
            "import ctypes\n"
            "buf = ctypes.create_string_buffer(128)\n"
            "rc = lib.algp_comm_unique_id(buf)\n"
            "rc2 = lib.algp_comm_unique_id(buf)\n"
            "print('RC', rc, rc2, lib.algp_version())\n" % REPO)
    env = dict(os.environ, ALGP_RCCL_PATH='/nonexistent/librccl_not_here.so')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert 'RC %d %d' % (_hip.ERR_HIP, _hip.ERR_HIP) in r.stdout, r.stdout


def test_bench_self_spawn_command_line(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE starts torch.distributed.run as a CHILD with the same arguments and
    relays exactly one JSON line (no GPU needed: the child is replaced by a stub that prints a line)."""
    import importlib.util
    import subprocess
    import sys
    import types
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(REPO, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, stdout=None, **kw):
        seen['cmd'] = cmd
        return types.SimpleNamespace(returncode=0, stdout=b'noise on stdout\n{"metric": "m", "value": 1.0, "n_gpus": 8}\n')
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '2', '--warmup', '1'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    rc = bench.self_spawn(types.SimpleNamespace(gpus=8))
    assert rc == 0
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '8', '--steps', '2', '--warmup', '1'] and cmd[-7].endswith('bench.py')


def test_a_build_without_test_hooks_exports_no_debug_symbol(tmp_path):
    """`make TEST_HOOKS=0` (-DALGP_TEST_HOOKS=0): the C-ABI translation units compiled that way (host side only: seconds each)
    define every product entry point and none of the nine algp_debug_* hooks (include/algp_hip.h, section "test hooks")."""
    import subprocess
    csrc = os.path.join(REPO, 'algp_amd', 'csrc')
    defined = set()
    for unit in sorted(f for f in os.listdir(csrc) if f.startswith('api') and f.endswith('.hip')):
        obj = tmp_path / (unit + '.o')
        r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O1', '-std=c++17', '-fPIC', '-DALGP_TEST_HOOKS=0', '-I' + csrc,
                            '--cuda-host-only', '-c', os.path.join(csrc, unit), '-o', str(obj)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (unit, r.stderr[-3000:])
        defined |= set(re.findall(r' T (algp_[a-z0-9_]+)', subprocess.run(['nm', str(obj)], capture_output=True, text=True).stdout))
    assert not [s for s in defined if s.startswith('algp_debug_')], defined
    decl = [s for s in _declared_symbols() if not s.startswith('algp_debug_')]
    assert sorted(defined) == decl, set(decl) ^ defined


def test_the_library_exports_the_c_abi_and_nothing_else():
    """csrc/exports.map: the dynamic symbol table holds the entry points of include/algp_hip.h only -- no C++ launcher, no
    template instantiation leaks out of the .so."""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', _hip.LIB_PATH], capture_output=True, text=True).stdout
    names = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert names == _declared_symbols(), set(names) ^ set(_declared_symbols())
