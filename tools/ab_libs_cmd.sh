# A/B of several builds of the library on the same box: tools/ab_libs_cmd.sh "<command>" lib1.so lib2.so ...
# (each build is copied over matcouply_amd/libmatcouply_hip.so in turn; the original is restored at the end)
cmd=$1; shift
cp matcouply_amd/libmatcouply_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  export lib
  cp "$lib" matcouply_amd/libmatcouply_hip.so
  echo "=== $lib"
  bash -c "$cmd"
done
cp /tmp/lib_keep.so matcouply_amd/libmatcouply_hip.so
