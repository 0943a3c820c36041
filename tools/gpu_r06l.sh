#!/bin/bash
# round-6 visit l: the final evidence once more (sph_api.hip changed after r06h: an empty record set answers its layer offsets with zeros)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/gpu_round.sh r06l tests bench kstats pmc pmcdf native dfgaps
