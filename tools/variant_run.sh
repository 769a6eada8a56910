# rebuild the library with extra compile flags, run a command, for each variant:  bash tools/variant_run.sh "<cmd>" "<flags1>" "<flags2>" ...
set -o pipefail
cmd="$1"; shift
for v in "$@"; do
  echo "=== variant [$v]"
  FD_EXTRA_HIPCC_FLAGS="$v" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
  bash -c "$cmd" || exit 1
done
