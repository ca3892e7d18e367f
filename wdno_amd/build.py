"""Build libwdno_hip.so (gfx950) in-tree with hipcc. No torch involvement: the library is a plain C-ABI .so."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libwdno_hip.so')
SOURCES = ['api.cpp', 'pointwise.hip', 'diffusion.hip', 'dwt.hip', 'conv.hip', 'conv_h3.hip', 'conv_h3d.hip', 'conv_wgrad_h3d.hip', 'norm.hip', 'attention.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math']


def hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)')


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'wdno_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    """Compile every translation unit and link wdno_amd/libwdno_hip.so. Returns the library path."""
    if not force and not _stale():
        return LIB
    cc = hipcc()
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.rsplit('.', 1)[0] + '.o')
        objs.append(obj)
        cmd = [cc] + FLAGS + ['-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
    cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv))
