python tools/exact_stack_prof.py c4 48 576 256 16 50 2>&1 | grep "us/iter"
python tools/exact_stack_prof.py c4 48 576 256 16 50 2>&1 | grep "us/iter"
md5sum matcouply_amd/libmatcouply_hip.so
