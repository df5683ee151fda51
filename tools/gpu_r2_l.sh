#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_search.sh "" "-DIA_EXP_COUNTERS=8" "-DIA_EXP_COUNTERS=64"
