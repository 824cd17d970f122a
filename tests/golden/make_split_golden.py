"""Golden capture of the reference's leave-one-out split (recsys/dataset.py:10-156) on a toy
ratings.dat / users.dat pair (run in the build container only; the reference is imported from
/root/reference and never shipped):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_split_golden.py

The toy files are generated here from a seed (users shuffled in the file, tied timestamps, movie ids
whose string order differs from their numeric order, fewer movies than the 500-item pool) and are
stored in the fixture as text lines beside every array the split produced."""
import os
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def toy_files(seed=5, n_users=14, n_movies=37):
    rng = np.random.RandomState(seed)
    movie_ids = np.sort(rng.choice(np.arange(1, 400), size=n_movies, replace=False))   # "9" > "10" as strings
    lines = []
    for u in range(1, n_users + 1):
        k = rng.randint(4, 12)
        seen = rng.choice(movie_ids, size=k, replace=False)
        ts = rng.randint(1000, 1010, size=k) * 100            # many ties per user
        for m, t in zip(seen, ts):
            lines.append((u, int(m), int(rng.randint(1, 6)), int(t)))
    order = rng.permutation(len(lines))                       # users not grouped in the file
    ratings = ["%d::%d::%d::%d" % lines[i] for i in order]
    ages = [1, 18, 25, 35, 45, 50, 56]
    users = ["%d::%s::%d::%d::%05d" % (u, "FM"[rng.randint(0, 2)], ages[rng.randint(0, 7)], rng.randint(0, 21), rng.randint(0, 99999))
             for u in range(1, n_users + 1)]
    return ratings, users


def coo(m):
    m = m.tocoo()
    order = np.lexsort((m.col, m.row))
    return {"rows": m.row[order].astype(np.int64), "cols": m.col[order].astype(np.int64), "vals": np.asarray(m.data[order]),
            "shape": np.array(m.shape, dtype=np.int64)}


def main():
    from recsys import dataset as ref_dataset
    from recsys.utils import path as ref_path
    ratings, users = toy_files()
    with tempfile.TemporaryDirectory() as tmp:
        rp, up, out = os.path.join(tmp, "ratings.dat"), os.path.join(tmp, "users.dat"), os.path.join(tmp, "ml1m.npz")
        open(rp, "w").write("\n".join(ratings) + "\n")
        open(up, "w").write("\n".join(users) + "\n")
        ref_path.ml1m_ratings, ref_path.ml1m_users, ref_path.ml1m_npz = rp, up, out
        ref_dataset.split_data_loo(n_test_codes=10)
        data = np.load(out, allow_pickle=True)["data"].item()      # written by this process a moment ago
    rec = {"ratings_lines": np.array(ratings), "users_lines": np.array(users), "n_test_codes": np.array(10, dtype=np.int64)}
    for k in ("rating_matrix", "implicit_matrix", "timestamp_matrix"):
        for kk, v in coo(data[k]).items():
            rec[f"{k}/{kk}"] = v
        rec[f"{k}/dtype"] = np.array(str(data[k].dtype))
    for k in ("test_codes", "test_true", "test_timestamp_rank", "negative_samples", "user_infos"):
        rec[k] = np.asarray(data[k])
        rec[f"{k}_dtype"] = np.array(str(np.asarray(data[k]).dtype))
    fn = os.path.join(HERE, "split.npz")
    np.savez_compressed(fn, **rec)
    print(fn, os.path.getsize(fn), {k: (v.shape, str(v.dtype)) for k, v in rec.items()})


if __name__ == "__main__":
    main()
