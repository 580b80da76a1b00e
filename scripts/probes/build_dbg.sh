# builds scripts/probes/variants/libffn_dbg.so from a /tmp/dbg/mlp.hip instrumented copy
set -e
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I fourier_feature_nets_amd/csrc -c /tmp/dbg/mlp.hip -o /tmp/dbg/mlp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/probes/variants/libffn_dbg.so $(ls fourier_feature_nets_amd/csrc/build/*.o | grep -v "/mlp.o") /tmp/dbg/mlp.o
echo built
