#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_field.sh "-DIA_ENC_NT=1" "-DIA_ENC_NT=0" 2>&1 | grep -E "===|uniform"
