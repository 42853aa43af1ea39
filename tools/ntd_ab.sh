# A/B of the dense single-output kernels per layer shape: k_ntl (MTLORA_NTD=0) vs k_ntd (=2: whenever eligible).  bash tools/ntd_ab.sh
S1="s1.qkv s1.fc1 s1.fc2 s2.qkv s2.fc1 s2.fc2 s3.qkv s3.fc1 s3.fc2 head0 head1 merge1"
S2="b0.fc2 b1.qkv b1.fc1 b1.fc2 b2.qkv b2.proj b2.fc1 b2.fc2 b3.qkv b3.fc1 b3.fc2"
for mode in "MTLORA_NTD=0" "MTLORA_NTD=1" "MTLORA_NTD=2"; do
  echo "== $mode"
  env $mode python tools/bench_linear.py --kinds --knt-only --shapes $S1 2>&1 | grep -E "fwd_outputs|plain_fwd|^[a-z0-9.]+ +M" | sed -E 's/k_pack.*//; s/k_tn.*//' | paste - - | awk '{print}' 
  env $mode python tools/bench_linear.py --kinds --knt-only --rs 128 --rt 128 --shapes $S2 2>&1 | grep -E "fwd_outputs|plain_fwd|^[a-z0-9.]+ +M" | sed -E 's/k_pack.*//; s/k_tn.*//' | paste - - 
done
