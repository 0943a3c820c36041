#!/bin/bash
# round-6 visit k: the final evidence once more (sph_api.hip changed after r06h: an empty record set answers its layer offsets with zeros)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/gpu_round.sh r06k tests bench kstats pmc pmcdf native dfgaps
