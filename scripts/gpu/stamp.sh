# run HERE (the build container) before a gpurun call whose artifacts name the profiled commit:
# .git does not travel to the GPU box, .git_head does (git-ignored)
git -C "$(dirname "$0")/../.." rev-parse --short HEAD > "$(dirname "$0")/../../.git_head"
if ! git -C "$(dirname "$0")/../.." diff --quiet HEAD -- fourier_feature_nets_amd include bench.py; then echo "-dirty" >> "$(dirname "$0")/../../.git_head"; fi
cat "$(dirname "$0")/../../.git_head"
