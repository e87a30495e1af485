export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python3 tools/exp/host_bound.py vlsac_halfcheetah_f256_b256 2>&1 | head -40
