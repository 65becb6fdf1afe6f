#!/bin/bash
# round 6, last session: SQ issue / wait counters of the FINAL k_maniac_decode (narrow supernodes, compact leaves, exact areas) on the
# workload round 3 measured (1024 x 1920x1080 with the group index: every wavefront slot busy for most of the launch), for
# the instructions-per-symbol comparison in DESIGN 4.1; then the same passes on 256 x 4K (the headline picture size).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/pmc_sq.sh r6_dense_1080p 1024 1920 1080 groups
bash tools/pmc_sq.sh r6_dense_4k 256 3840 2160 groups
