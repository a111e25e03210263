#!/bin/bash
# register / scratch / LDS use of every kernel of one source file:  bash tools/kres.sh <file.hip> [extra flags]
cd "$(dirname "$0")/../msu-latentafis_amd/csrc"
F=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -Rpass-analysis=kernel-resource-usage -c $F -o /tmp/kres_$$.o 2>&1 | \
  awk '/Function Name/{n=$(NF-1)} / VGPRs:/{v=$(NF-1)} /AGPRs:/{a=$(NF-1)} /ScratchSize/{s=$(NF-1)} /Occupancy/{o=$(NF-1)} /LDS Size/{l=$(NF-1); printf "%-90s vgpr %s agpr %s scratch %s occ %s lds %s\n", substr(n,1,90), v, a, s, o, l}'
rm -f /tmp/kres_$$.o
