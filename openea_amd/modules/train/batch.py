"""Batching, negative sampling and neighbour search (mirror of openea/modules/train/batch.py).

Two layers:

* the reference's function signatures (python lists / dicts in and out) -- a drop-in for code
  that calls ``bat.generate_*`` -- which convert at the edge and run the HIP kernels;
* device-resident objects (``TripleSampler``, ``EpochBatches``) used by ``BasicModel`` so that
  nothing crosses PCIe inside an epoch.

There is no CPU implementation here: without libopenea_hip.so / a GPU every sampling or search
call raises ``OpenEAHipError``.
"""
import os

import numpy as np
import torch

from .. import _seed
from ... import ops

# ----------------------------------------------------------------------------------------------
# positive batching (pure index arithmetic, batch.py:17-22, 48-57)
# ----------------------------------------------------------------------------------------------


def batch_sizes(n1, n2, batch_size):
    """batch.py:18-19: b1 = int(n1 / (n1 + n2) * B) in python float arithmetic, b2 = B - b1."""
    b1 = int(n1 / (n1 + n2) * batch_size)
    return b1, batch_size - b1


def generate_pos_triples(triples, batch_size, step, is_fixed_size=False):
    """batch.py:48-57: the contiguous slice [step*b, step*b + b) of the (shuffled) list."""
    start = step * batch_size
    end = min(start + batch_size, len(triples))
    pos_batch = triples[start:end]
    if is_fixed_size and len(pos_batch) < batch_size:
        pos_batch = pos_batch + triples[:batch_size - len(pos_batch)]
    return pos_batch


def generate_pos_batch(triple_list1, triple_list2, batch_size, step):
    """batch.py:17-22."""
    b1, b2 = batch_sizes(len(triple_list1), len(triple_list2), batch_size)
    return generate_pos_triples(triple_list1, b1, step) + generate_pos_triples(triple_list2, b2, step)


def generate_pos_batch_queue(triple_list1, triple_list2, batch_size, steps, out_queue):
    """batch.py:11-14 (kept for callers that still run a producer; the model does not)."""
    for step in steps:
        out_queue.put(generate_pos_batch(triple_list1, triple_list2, batch_size, step))


# ----------------------------------------------------------------------------------------------
# device-resident sampler
# ----------------------------------------------------------------------------------------------


class TripleSampler:
    """Device state replacing (all_triples_set, entities_list, neighbor) of
    generate_neg_triples_fast (batch.py:89-119) for ONE KG."""

    def __init__(self, triples_set, entities_list, num_entities_total=None, dev=None):
        dev = dev or ops.device()
        tri = np.asarray(sorted(triples_set) if not isinstance(triples_set, np.ndarray) else triples_set,
                         dtype=np.int32).reshape(-1, 3)
        self.n_triples = len(tri)
        self.entity_list = ops.to_ids(np.asarray(entities_list, np.int32), dev)
        n_total = int(num_entities_total if num_entities_total is not None else (max(entities_list) + 1))
        ent_pos = np.full(n_total, -1, np.int32)
        ent_pos[np.asarray(entities_list, np.int64)] = np.arange(len(entities_list), dtype=np.int32)
        self.ent_pos = ops.to_ids(ent_pos, dev)
        tri_dev = ops.to_ids(tri, dev)
        self.table = ops.tripleset_build(tri_dev)
        # the "certainly absent" bit array in front of the key table (csrc/sampler.hip; OEA_SAMPLER_FILTER=0: probes only)
        self.filter = ops.tripleset_filter(tri_dev, self.table.numel()) if os.environ.get("OEA_SAMPLER_FILTER", "1")[:1] != "0" else None
        self.nbr = None                       # int32 [N, k] entity ids; row = position in entity_list
        self.nbr_pos = None
        # the membership keys pack (head 24 bits | relation 16 bits | tail 24 bits): larger ids would alias silently
        if len(tri) and (int(tri[:, [0, 2]].max()) >= 1 << 24 or int(tri[:, 1].max()) >= 1 << 16 or int(tri.min()) < 0):
            raise ops.OpenEAHipError("triple ids out of range for the packed membership keys "
                                     "(entity ids < 16,777,216, relation ids < 65,536)")
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)

    def set_neighbours(self, nbr, nbr_pos=None):
        """nbr: device int32 [rows, k] or None; rows in entity_list order unless nbr_pos (device int32
        [num_entities_total]: entity id -> row, -1 = no list: the whole entity list is the candidate set) is given."""
        self.nbr = nbr
        self.nbr_pos = nbr_pos

    def side(self):
        """this KG's state packed for ops.sample_negatives_pair."""
        pos = self.nbr_pos if (self.nbr is not None and getattr(self, "nbr_pos", None) is not None) else self.ent_pos
        return ops.sampler_side(self.table, self.entity_list, pos, self.nbr, getattr(self, "filter", None))

    def sample(self, pos, k, seed, step, pos_offset=0, out=None, max_try=10):
        """pos: device int32 [n,3] -> device int32 [n*k, 3]."""
        ent_pos = self.nbr_pos if (self.nbr is not None and getattr(self, "nbr_pos", None) is not None) else self.ent_pos
        out, _ = ops.sample_negatives(pos, k, self.table, self.entity_list, ent_pos, self.nbr, seed=seed,
                                      step=step, pos_offset=pos_offset, max_try=max_try, out=out, err_flag=self.err)
        return out

    def check(self):
        if int(self.err.item()) != 0:
            raise ValueError("Sample larger than population or is negative")   # random.sample's error


class EpochBatches:
    """All positive batches of one epoch, resident in HBM.  batch `step` is the concatenation of
    KG1's slice [step*b1, ...) and KG2's slice [step*b2, ...) exactly as generate_pos_batch
    builds it (batch.py:17-22); the arrays are re-uploaded after each epoch's shuffle
    (basic_model.py:234-235)."""

    def __init__(self, triples1, triples2, batch_size, dev=None):
        self.dev = dev or ops.device()
        self.t1 = np.asarray(triples1, np.int32).reshape(-1, 3)
        self.t2 = np.asarray(triples2, np.int32).reshape(-1, 3)
        self.b1, self.b2 = batch_sizes(len(self.t1), len(self.t2), batch_size)
        self.upload()

    def upload(self):
        """Lay the epoch out batch by batch -- [step0: KG1 slice | KG2 slice][step1: ...] -- so that
        batch `step` is ONE contiguous device slice (no per-step concatenation).  The layout is a
        fixed gather map over cat(list1, list2); an epoch's shuffle only permutes its input."""
        n1, n2, b1, b2 = len(self.t1), len(self.t2), self.b1, self.b2
        # basic_model.py:255: triple_steps = ceil((n1 + n2) / batch_size).  b1 rounds down, so the last steps hold
        # short (or empty) slices and a few tail triples of the list with the rounded-down share are not visited in
        # this epoch -- exactly as in the reference (the lists are reshuffled every epoch).
        steps = int(np.ceil((n1 + n2) / max(b1 + b2, 1)))
        slot, offsets, splits = [], [0], []
        for s in range(steps):
            i1 = np.arange(s * b1, min(s * b1 + b1, n1)) if b1 else np.zeros(0, np.int64)
            i2 = np.arange(s * b2, min(s * b2 + b2, n2)) + n1 if b2 else np.zeros(0, np.int64)
            slot += [i1, i2]
            splits.append(len(i1))
            offsets.append(offsets[-1] + len(i1) + len(i2))
        self.offsets = np.asarray(offsets, np.int64)
        self.splits = np.asarray(splits, np.int64)
        slot = np.concatenate(slot).astype(np.int64) if slot else np.zeros(0, np.int64)
        self.slot = torch.from_numpy(slot).to(self.dev)
        self.tall = ops.to_ids(np.concatenate([self.t1, self.t2]), self.dev)      # [n1+n2, 3]
        self.dall = self.tall[self.slot].contiguous()
        self.n1, self.n2 = n1, n2

    def shuffle(self, gen=None, into_next=False):
        """random.shuffle of both lists (basic_model.py:234-235) + the fixed batch layout gather in ONE library call
        (oea_epoch_layout: Philox keys, one radix sort, one gather; torch.randperm x 2 + cat + two index gathers were 17 launches
        and 0.66 ms per epoch at the 100K shape).  Every epoch draws a FRESH permutation of the lists as loaded -- the
        distribution of shuffling last epoch's order.  The generator only seeds the stream (its initial seed; the epoch counter
        is the Philox counter), so two processes with the same seed lay out the same epochs.  No host round trip.
        into_next: the new layout goes to a second buffer (`dall_next`) that `swap()` makes current -- the next epoch is laid
        out on a side stream while the current one is still being consumed."""
        seed = int(gen.initial_seed()) if gen is not None else 0
        self._shuffles = getattr(self, "_shuffles", 0) + 1
        if into_next and getattr(self, "dall_next", None) is None:
            self.dall_next = torch.empty_like(self.dall)
        out = self.dall_next if into_next else self.dall
        self._layout_ws = ops.epoch_layout(self.tall, self.n1, self.n2, self.slot, seed, self._shuffles, out,
                                           getattr(self, "_layout_ws", None))

    def swap(self):
        self.dall, self.dall_next = self.dall_next, self.dall

    def pos(self, step):
        """-> (device [n,3] batch, n_split): rows [0, n_split) come from KG1 (batch.py:45)."""
        if step + 1 >= len(self.offsets):
            return self.dall[:0], 0
        return self.dall[int(self.offsets[step]): int(self.offsets[step + 1])], int(self.splits[step])


# ----------------------------------------------------------------------------------------------
# reference-signature wrappers
# ----------------------------------------------------------------------------------------------
_sampler_cache = {}


def _fingerprint(all_triples_set, entities_list):
    """cheap content fingerprint: sizes + a strided sample of ~256 triples (itertools.islice: the walk over the set is a
    C loop, no sort, no per-element python work) + the ends of the entity list"""
    from itertools import islice
    n, m = len(all_triples_set), len(entities_list)
    sample = tuple(islice(all_triples_set, 0, None, max(1, n // 256)))
    return (n, hash(sample), m, entities_list[0] if m else -1, entities_list[-1] if m else -1)


def invalidate_sampler_cache():
    """drop the cached device samplers: call after mutating a triple set or an entity list IN PLACE (see _cached_sampler)"""
    _sampler_cache.clear()


def _cached_sampler(all_triples_set, entities_list):
    """one device sampler per (triple set, entity list).  The key is (id, id): O(1) per call; the stored fingerprint
    (sizes + a strided sample of ~256 triples + the ends of the entity list) is compared on every hit, which catches a
    recycled id() and any mutation that changes a size.  It is NOT a content hash: a set mutated in place with its size
    kept (one triple removed, another added) or an entity list edited in the middle almost always passes it and would get
    the stale device table -- negatives filtered against the old set.  The reference never mutates these containers
    (kgs.py builds them once); a caller that does must call invalidate_sampler_cache().  The sorted [n, 3] array is only
    built on a miss."""
    key = (id(all_triples_set), id(entities_list))
    fp = _fingerprint(all_triples_set, entities_list)
    hit = _sampler_cache.get(key)
    if hit is not None and hit[0] == fp:
        return hit[1]
    if len(_sampler_cache) > 8:
        _sampler_cache.clear()
    tri = np.asarray(sorted(all_triples_set), dtype=np.int32).reshape(-1, 3)
    ents = np.asarray(entities_list, dtype=np.int32)
    s = TripleSampler(tri, ents)
    _sampler_cache[key] = (fp, s)
    return s


def generate_neg_triples_fast(pos_batch, all_triples_set, entities_list, neg_triples_num, neighbor=None, max_try=10):
    """Drop-in for batch.py:89-119: same arguments, returns a list of (h, r, t) tuples of length
    neg_triples_num * len(pos_batch) (negatives of positive p at [p*k, (p+1)*k)).
    The draws come from the device Philox stream (seed: openea_amd.modules.set_seed)."""
    if len(pos_batch) == 0:
        return []
    sampler = _cached_sampler(all_triples_set, entities_list)
    if neighbor:
        # neighbor.get(e, entities_list) (batch.py:96-97): entities without a list draw from the WHOLE entity list --
        # ent_pos = -1 makes the kernel fall back to it
        k_n = len(next(iter(neighbor.values())))
        have = [e for e in entities_list if e in neighbor]
        nbr = np.asarray([neighbor[e] for e in have], np.int32).reshape(len(have), k_n)
        ent_pos = np.full(sampler.ent_pos.numel(), -1, np.int32)
        ent_pos[np.asarray(have, np.int64)] = np.arange(len(have), dtype=np.int32)
        sampler.set_neighbours(ops.to_ids(nbr), ops.to_ids(ent_pos))
    else:
        sampler.set_neighbours(None)
    pos = ops.to_ids(np.asarray(pos_batch, np.int32).reshape(-1, 3))
    out = sampler.sample(pos, neg_triples_num, _seed.get_seed(), _seed.next_call(), max_try=max_try)
    sampler.check()
    neg = out.cpu().numpy()
    assert len(neg) == neg_triples_num * len(pos_batch)
    return [tuple(int(x) for x in row) for row in neg]


def generate_neg_triples(pos_batch, all_triples_set, entities_list, neg_triples_num, neighbor=None, max_try=10):
    """batch.py:60-86 draws one candidate per negative; its accepted negatives have the same
    distribution as one round of the `fast` variant per needed sample, which is what runs."""
    return generate_neg_triples_fast(pos_batch, all_triples_set, entities_list, neg_triples_num, neighbor, max_try)


def generate_relation_triple_batch(triple_list1, triple_list2, triple_set1, triple_set2, entity_list1, entity_list2,
                                   batch_size, step, neighbor1, neighbor2, neg_triples_num):
    """batch.py:36-45."""
    b1, b2 = batch_sizes(len(triple_list1), len(triple_list2), batch_size)
    pos_batch1 = generate_pos_triples(triple_list1, b1, step)
    pos_batch2 = generate_pos_triples(triple_list2, b2, step)
    neg_batch1 = generate_neg_triples_fast(pos_batch1, triple_set1, entity_list1, neg_triples_num, neighbor=neighbor1)
    neg_batch2 = generate_neg_triples_fast(pos_batch2, triple_set2, entity_list2, neg_triples_num, neighbor=neighbor2)
    return pos_batch1 + pos_batch2, neg_batch1 + neg_batch2


def generate_relation_triple_batch_queue(triple_list1, triple_list2, triple_set1, triple_set2, entity_list1,
                                         entity_list2, batch_size, steps, out_queue, neighbor1, neighbor2,
                                         neg_triples_num):
    """batch.py:25-33."""
    for step in steps:
        out_queue.put(generate_relation_triple_batch(triple_list1, triple_list2, triple_set1, triple_set2,
                                                     entity_list1, entity_list2, batch_size, step, neighbor1,
                                                     neighbor2, neg_triples_num))


# ----------------------------------------------------------------------------------------------
# neighbour search (batch.py:122-165)
# ----------------------------------------------------------------------------------------------


def neighbours_device(embeds, dim, entity_ids, k):
    """embeds: device [N, ld] rows in entity_list order; entity_ids: device int32 [N].
    -> device int32 [N, k] neighbour ENTITY IDS (ascending candidate position per row)."""
    return ops.topk_inner(embeds, embeds, dim, k, id_map=entity_ids)


def find_neighbours(frags, entity_list, sub_embed, embed, k):
    """batch.py:157-165: {frags[i] -> list of the k nearest entity ids (unordered set in the
    reference; ascending candidate position here)}."""
    d = embed.shape[1]
    idx = ops.topk_inner(ops.to_table(sub_embed), ops.to_table(embed), d, k,
                         id_map=ops.to_ids(np.asarray(entity_list, np.int32))).cpu().numpy()
    return {frags[i]: idx[i].tolist() for i in range(len(frags))}


def generate_neighbours_single_thread(entity_embeds, entity_list, neighbors_num, threads_num):
    """batch.py:145-154.  `threads_num` only fragmented the host matmul; one device call here."""
    ents = np.asarray(entity_list)
    return find_neighbours(ents.tolist(), ents, entity_embeds, entity_embeds, neighbors_num)


def generate_neighbours(entity_embeds, entity_list, neighbors_num, threads_num):
    """batch.py:122-142."""
    return generate_neighbours_single_thread(entity_embeds, entity_list, neighbors_num, threads_num)
