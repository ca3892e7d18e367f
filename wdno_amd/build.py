"""Build libwdno_hip.so (gfx950) in-tree with hipcc. No torch involvement: the library is a plain C-ABI .so."""
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libwdno_hip.so')
SOURCES = ['api.cpp', 'pointwise.hip', 'diffusion.hip', 'pack.hip', 'dwt.hip', 'conv.hip', 'conv_h3.hip', 'conv_h3d.hip', 'conv_h3t.hip', 'conv_wgrad_h3d.hip', 'norm.hip', 'attention.hip', 'attn_fused.hip', 'attn_fused48.hip', 'attn_fused_wide.hip', 'attn_fused_bwd.hip', 'linattn_fused.hip', 'linattn_fused_wide.hip', 'linattn_fused_bwd.hip', 'linear_rows.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math']
RESOURCES = os.path.join(HERE, 'build', 'kernel_resources.json')     # per-kernel registers / scratch from the last build


def _parse_resources(text):
    """{kernel: {vgprs, agprs, scratch_bytes, vgpr_spill, occupancy}} from hipcc -Rpass-analysis=kernel-resource-usage."""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r'remark: (?:\s*)Function Name: (\S+)', line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (('vgprs', r'VGPRs: (\d+)'), ('agprs', r'AGPRs: (\d+)'), ('scratch_bytes', r'ScratchSize \[bytes/lane\]: (\d+)'),
                         ('vgpr_spill', r'VGPRs Spill: (\d+)'), ('occupancy', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds_bytes', r'LDS Size \[bytes/block\]: (\d+)')):
            m = re.search(r'remark:\s+' + pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


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
        cmd = [cc] + FLAGS + ['-Rpass-analysis=kernel-resource-usage', '-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    resources = {}
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
        resources.update(_parse_resources(out))
    with open(RESOURCES, 'w') as f:
        json.dump(resources, f, indent=1, sort_keys=True)
    for k, v in sorted(resources.items()):          # a kernel that spills runs several times slower: never let it pass silently
        if v.get('vgpr_spill', 0) or v.get('scratch_bytes', 0):
            print(f'WARNING: {k} spills ({v.get("vgpr_spill", 0)} VGPRs, {v.get("scratch_bytes", 0)} B scratch per lane)', flush=True)
    cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv))
