# Build libmatcouply_hip.so of another git revision into build_ab/<name>.so for same-box A/B runs (tools/ab_lib.sh):
#   bash tools/build_rev_lib.sh <git-rev> <name>
set -e
REV=$1; NAME=$2; D=build_ab/src_$NAME
rm -rf $D; mkdir -p $D/matcouply_amd/csrc $D/include
for f in $(git ls-tree --name-only $REV matcouply_amd/csrc/); do git show $REV:$f > $D/$f; done
git show $REV:include/matcouply_hip.h > $D/include/matcouply_hip.h
OBJS=""
# per-file options of THAT revision's build (matcouply_amd/_build.py: EXTRA_FLAGS), none before they existed
git show $REV:matcouply_amd/_build.py > $D/_build_rev.py
for src in $D/matcouply_amd/csrc/*.hip; do
  o=${src%.hip}.o
  extra=$(python3 - "$D/_build_rev.py" "$(basename $src)" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
m = re.search(r"^EXTRA_FLAGS = (\{.*\})$", txt, re.M)
print(" ".join(eval(m.group(1)).get(sys.argv[2], [])) if m else "")
PY
)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c $src -o $o &
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/$NAME.so $OBJS
ls -la build_ab/$NAME.so
