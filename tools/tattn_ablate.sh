# timing ablations of csrc/attn_fused_bwd.hip: a build of the library with -DWDNO_TB_ABLATIONS (in a scratch copy), then tools/bench_tattn.py per variant
set -e
R=$GRAFT_REPO_ROOT
S=/tmp/abl_repo
rm -rf $S && cp -r $R $S && cd $S
python - <<'P'
from wdno_amd import build
build.FLAGS.append('-DWDNO_TB_ABLATIONS')
build.build_library(force=True, verbose=False)
P
for a in 0 1 2 3 4 5; do echo "ABL=$a"; WDNO_TB_ABLATE=$a timeout 100 python tools/bench_tattn.py 2>&1 | grep "with gradients" | cut -c1-100; done
