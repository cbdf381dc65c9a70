#!/bin/bash
# Commit of the working tree, with "+dirty" when tracked files differ from it: pass it to the GPU box (which has no .git)
# as TPL_GIT_HEAD so that a profile's summary.json can say what it measured.
cd "$(dirname "$0")/.."
h=$(git rev-parse --short=12 HEAD)
[ -n "$(git status --porcelain --untracked-files=no)" ] && h="$h+dirty"
echo "$h"
