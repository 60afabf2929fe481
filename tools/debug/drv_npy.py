import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from mirage_amd import run_pretraining as R
rng = np.random.default_rng(0)
root = "/tmp/mrgdata"
for d, (hw, hi) in {"bscan": (64, 256), "slo": (64, 256), "bscanlayermap": (128, 13)}.items():
    os.makedirs(os.path.join(root, d), exist_ok=True)
    for i in range(8):
        np.save(os.path.join(root, d, f"s{i:03d}.npy"), rng.integers(0, hi, (hw, hw), dtype=np.uint8))
argv = ["--data_path", root, "--in_domains", "bscan-slo-bscanlayermap", "--out_domains", "bscan-slo-bscanlayermap",
        "--input_size", "64", "--patch_size", "8", "--batch_size", "4", "--num_encoded_tokens", "24", "--epochs", "2",
        "--warmup_epochs", "1", "--save_ckpt_freq", "1", "--base_output_dir", "/tmp/mrgout", "--print_freq", "1", "--no_graph"] + sys.argv[1:]
R.main(R.get_args(argv))
print("OK")
