"""Row-by-row comparison of two maps that were GROWN, not given: where every row comes from.

Growth appends rows (gaussian_map.py:294-468) and prune is a stable compaction (:234-246), so an ORIGIN id
(keyframe << 32 | index among the rows that keyframe added) rides along from outside when ``add_gaussians`` and
``prune`` are wrapped.  tests/golden/mapper_loop.pt holds the reference's ids (``final.origin``) and the rows each
keyframe spawned (``history[k].added_means``); ``RowOrigins`` keeps the same for a trainer of this repository, and
``common_rows`` pairs the two maps' final rows through the spawned rows' positions (one surfel per 2 cm voxel: a spawned
row is identified by where it was put) - so the final parameters are compared ALWAYS, also when a pixel on one of
add_gaussians' thresholds made the two maps differ by a row."""
import torch


class RowOrigins:
    def __init__(self, trainer):
        self.tr, self.k = trainer, 0
        self.origin = torch.zeros(0, dtype=torch.int64)
        self.added = []                 # per keyframe: the rows it spawned, as spawned
        self.pruned = []                # per prune call: rows deleted
        add0, prune0 = trainer.add_gaussians, trainer.prune

        def add(frame):
            n0 = trainer.means.shape[0]
            out = add0(frame)
            n1 = trainer.means.shape[0]
            self.added.append(trainer.means[n0:n1].detach().cpu().clone())
            self.origin = torch.cat([self.origin, (self.k << 32) + torch.arange(n1 - n0, dtype=torch.int64)])
            self.k += 1
            return out

        def prune(mask):
            # the rule of gaussian_map.py:234-246: the caller's mask OR opacity < 0.1
            gone = mask.bool().cpu() | (torch.sigmoid(trainer.opacities.detach()).cpu() < 0.1)
            out = prune0(mask)
            assert int((~gone).sum()) == trainer.means.shape[0], "prune kept other rows than mask | opacity < 0.1 says"
            self.origin = self.origin[~gone]
            self.pruned.append(int(gone.sum()))
            return out

        trainer.add_gaussians, trainer.prune = add, prune


def common_rows(ref_added, ref_origin, my_added, my_origin, tol=1e-4):
    """-> (index into the reference's final rows, index into mine, per-keyframe counts) of the rows BOTH maps hold:
    spawned by the same keyframe at the same place (within ``tol`` metres; spawned rows are >= a pixel footprint apart)
    and still present in both."""
    ref_pos = {int(o): i for i, o in enumerate(ref_origin.tolist())}
    my_pos = {int(o): i for i, o in enumerate(my_origin.tolist())}
    ri, mi, stats = [], [], []
    for k, (a, b) in enumerate(zip(ref_added, my_added)):
        a = torch.zeros(0, 3) if a is None else a.float().cpu()
        b = b.float().cpu()
        paired = 0
        if a.shape[0] and b.shape[0]:
            d = torch.cdist(a.double(), b.double())
            nearest_b = d.argmin(1)
            nearest_a = d.argmin(0)
            for j in range(a.shape[0]):
                jb = int(nearest_b[j])
                if int(nearest_a[jb]) == j and float(d[j, jb]) < tol:
                    paired += 1
                    ro, mo = (k << 32) + j, (k << 32) + jb
                    if ro in ref_pos and mo in my_pos:
                        ri.append(ref_pos[ro]); mi.append(my_pos[mo])
        stats.append(dict(keyframe=k, spawned_ref=int(a.shape[0]), spawned_mine=int(b.shape[0]), same_place=paired))
    return torch.tensor(ri, dtype=torch.long), torch.tensor(mi, dtype=torch.long), stats
