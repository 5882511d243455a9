DISYOLO_LIB=$GRAFT_REPO_ROOT/dis-yolo_amd/libdisyolo_probe.so python tools/probe_halo.py 2>&1 | grep -v amdgpu.ids
