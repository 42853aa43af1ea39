# A/B of the factor-gradient kernels per layer shape: tiled k_tn (MTLORA_SP_TN=0) vs wave-streaming k_sp_tn (=2: whenever eligible).  bash tools/tn_ab.sh
S1="s0.qkv s0.fc2 s0.projT s0.fc1T s0.fc2T s1.qkv s1.fc2 s1.fc1T s1.fc2T s2.qkv s2.fc2"
S2="b0.qkv b0.fc2 b1.fc1 b2.qkv b2.fc1 b2.fc2 b2.fc1T b3.fc1"
for mode in "MTLORA_SP_TN=0" "MTLORA_SP_TN=2"; do
  echo "== $mode"
  env $mode python tools/bench_linear.py --kinds --knt-only --shapes $S1 2>&1 | grep -o "k_tn:dA_dB [0-9.]*us" | tr '\n' ' '; echo
  env $mode python tools/bench_linear.py --kinds --knt-only --rs 128 --rt 128 --shapes $S2 2>&1 | grep -o "k_tn:dA_dB [0-9.]*us" | tr '\n' ' '; echo
done
