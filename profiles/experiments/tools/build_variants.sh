#!/bin/bash
# Build experimental variants of the library into build/exp/<name>.so:  tools/build_variants.sh name "-DFLAGS" [name flags ...]
set -e
cd "$(dirname "$0")/../spliser_amd/csrc"
mkdir -p ../../build/exp
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  tmp=$(mktemp -d)
  defs=$(for f in $flags; do case $f in -D*) echo -n "$f ";; esac; done)   # (the host-only files take the -D flags too)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -c spl_kernels.hip -o $tmp/k.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c spl_inflate.hip -o $tmp/z.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c spl_devpack.hip -o $tmp/dp.o
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread $flags -c spl_capi.cpp -o $tmp/c.o
  g++ -O3 -std=c++17 -fPIC -pthread $defs -c bam_reader.cpp -o $tmp/b.o
  g++ -O3 -std=c++17 -fPIC -pthread $defs -c spl_host.cpp -o $tmp/h.o
  g++ -O3 -std=c++17 -fPIC -pthread $defs -c spl_pack.cpp -o $tmp/p.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build/exp/$name.so $tmp/k.o $tmp/z.o $tmp/dp.o $tmp/c.o $tmp/b.o $tmp/h.o $tmp/p.o -lz -ldl -lpthread
  rm -rf $tmp
  echo built build/exp/$name.so
done
