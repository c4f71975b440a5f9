#!/bin/bash
# A/B two builds on ONE GPU box (box-to-box and thermal spread is +-1.5 %, more than most changes):
# check an older commit out into ./_ab_old, build its library there, and run the same command in
# both trees back to back inside a single gpurun call.
#   tools/ab_worktree.sh <commit> '<command run in each tree>' [rounds]
# e.g. tools/ab_worktree.sh HEAD~3 'python bench.py --steps 10 --warmup 3 --no-cpu-baseline | cut -c1-160' 2
set -eu
commit=$1; cmd=$2; rounds=${3:-2}
root=$(git rev-parse --show-toplevel)
cd "$root"
git worktree remove --force _ab_old 2>/dev/null || true
git worktree add -f _ab_old "$commit" -q
( cd _ab_old && python -m scalellm_amd.build | tail -1 && make -C oracle >/dev/null )
loop="for i in \$(seq $rounds); do for d in . _ab_old; do echo \"== \$d\"; (cd \$d && $cmd); done; done"
gpurun --timeout 900 -- "$loop" 2>&1 | tail -$((rounds * 6 + 4))
git worktree remove --force _ab_old
git worktree prune
