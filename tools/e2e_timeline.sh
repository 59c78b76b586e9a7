#!/bin/bash
# Copy / kernel timeline of `process` end to end (run on the GPU box): rocprofv3 kernel + memory-copy trace of tools/e2e_profile.py,
# summarised by tools/e2e_timeline.py (H2D rate of the staging ring's pieces as they ran, how much of the copy time had a kernel
# running beside it, idle gaps).    usage: tools/e2e_timeline.sh <tag> <workload> [e2e_profile.py arguments...]
TAG=$1; W=$2; shift 2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out /tmp/wl
cd $R && python3 tools/e2e_profile.py $W --runs 2 "$@" > $R/gpurun_out/${TAG}_e2e_${W}_plain.log 2>&1     # writes the files, warms the page cache
cd /tmp && rm -rf /tmp/tl_$W
# (two calls under the tracer, a second of sleep between them: the summary is of the second, a call of a process that has its
#  HIP context and its device memory -- the first one's is in the trace directory for whoever wants it)
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$W -- python3 $R/tools/e2e_profile.py $W --runs 2 --pause 1 "$@" > $R/gpurun_out/${TAG}_e2e_${W}_traced.log 2>&1
python3 $R/tools/e2e_timeline.py /tmp/tl_$W --last-call > $R/gpurun_out/${TAG}_e2e_${W}_timeline.txt 2>&1
tail -3 $R/gpurun_out/${TAG}_e2e_${W}_plain.log; tail -40 $R/gpurun_out/${TAG}_e2e_${W}_timeline.txt
