"""Writes tiny files in the KuaiRec on-disk layout (small_matrix.csv, item_categories.json, photo_mean_duration.json)."""
import json
import os

import pandas as pd


def write_kuairec_files(root, log_user, log_photo, log_ratio, list_feat, durations):
    os.makedirs(root, exist_ok=True)
    pd.DataFrame({"user_id": log_user, "photo_id": log_photo, "play_duration": 1, "watch_ratio": log_ratio}).to_csv(
        os.path.join(root, "small_matrix.csv"), index=False)
    with open(os.path.join(root, "item_categories.json"), "w") as fh:
        json.dump({str(i): {"feature_index": [int(c) for c in f if c >= 0]} for i, f in enumerate(list_feat)}, fh)
    with open(os.path.join(root, "photo_mean_duration.json"), "w") as fh:
        json.dump({str(i): float(d) for i, d in enumerate(durations)}, fh)
