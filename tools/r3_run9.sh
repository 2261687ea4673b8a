set -u
cd "$GRAFT_REPO_ROOT"
bash tools/prof_r3.sh || exit 1
bash tools/prof.sh || exit 1
echo profiles done
