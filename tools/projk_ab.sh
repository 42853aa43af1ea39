# A/B of the P / Q passes with large K R: tiled k_nt (MTLORA_SP_PROJK=0) vs k_sp_projk (=1 auto).  bash tools/projk_ab.sh
S1="s1.fc2 s2.qkv s2.fc1 s2.fc2 s3.qkv s3.fc1 s3.fc2"
S2="b1.fc1 b1.fc2 b2.qkv b2.proj b2.fc1 b2.fc2 b2.fc1T b2.fc2T b3.qkv b3.fc1 b3.fc2"
for mode in "MTLORA_SP_PROJK=0" "MTLORA_SP_PROJK=1"; do
  echo "== $mode"
  env $mode python tools/bench_linear.py --kinds --knt-only --shapes $S1 2>&1 | grep -E "lowrank|^[a-z0-9.]+ +M" | grep -o "k_nt:fwd_lowrank_P [0-9.]*us\|k_nt:bwd_lowrank_Q [0-9.]*us\|^[a-zA-Z0-9.]* " | paste - - - | tr '\n' ';'; echo
  env $mode python tools/bench_linear.py --kinds --knt-only --rs 128 --rt 128 --shapes $S2 2>&1 | grep -E "lowrank|^[a-z0-9.]+ +M" | grep -o "k_nt:fwd_lowrank_P [0-9.]*us\|k_nt:bwd_lowrank_Q [0-9.]*us\|^[a-zA-Z0-9.]* " | paste - - - | tr '\n' ';'; echo
done
