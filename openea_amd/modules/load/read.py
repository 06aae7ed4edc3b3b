"""Dataset reader, id assignment and writers (mirror of openea/modules/load/read.py).

These functions define the id layout the hot path consumes (KG1 even / KG2 odd ids in
descending-frequency order, read.py:64-92) and the on-disk formats it produces
(``ent_embeds.npy`` + id tsv files, read.py:318-366).  One-off host work: not accelerated
(SURVEY 8f rank 2), kept byte-compatible.
"""
import os
from collections import Counter

import numpy as np


def load_embeddings(file_name):
    return np.load(file_name) if os.path.exists(file_name) else None


def sort_elements(triples, elements_set):
    """read.py:12-29: frequency of each element over all three triple positions; order by
    (frequency, key) descending."""
    freq = Counter()
    for s, p, o in triples:
        if s in elements_set:
            freq[s] += 1
        if p in elements_set:
            freq[p] += 1
        if o in elements_set:
            freq[o] += 1
    dic = {e: freq.get(e, 0) for e in elements_set}
    ordered = [k for k, _ in sorted(dic.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)]
    return ordered, dic


def generate_mapping_id(kg1_triples, kg1_elements, kg2_triples, kg2_elements, ordered=True):
    """read.py:64-92."""
    ids1, ids2 = {}, {}
    if ordered:
        o1, _ = sort_elements(kg1_triples, kg1_elements)
        o2, _ = sort_elements(kg2_triples, kg2_elements)
        n1, n2 = len(o1), len(o2)
        both = min(n1, n2)
        for i in range(both):
            ids1[o1[i]] = 2 * i
            ids2[o2[i]] = 2 * i + 1
        for i in range(both, n2):          # KG1 exhausted
            ids2[o2[i]] = n1 * 2 + (i - n1)
        for i in range(both, n1):          # KG2 exhausted
            ids1[o1[i]] = n2 * 2 + (i - n2)
    else:
        index = 0
        for ele in kg1_elements:
            if ele not in ids1:
                ids1[ele] = index
                index += 1
        for ele in kg2_elements:
            if ele not in ids2:
                ids2[ele] = index
                index += 1
    assert len(ids1) == len(set(kg1_elements))
    assert len(ids2) == len(set(kg2_elements))
    return ids1, ids2


def generate_sharing_id(train_links, kg1_triples, kg1_elements, kg2_triples, kg2_elements, ordered=True):
    """read.py:32-61: seed-linked KG2 elements share the id of their KG1 counterpart."""
    ids1, ids2 = {}, {}
    if ordered:
        linked = {y: x for x, y in train_links}
        kg2_linked = [y for _, y in train_links]
        kg2_unlinked = set(kg2_elements) - set(kg2_linked)
        ids1, ids2 = generate_mapping_id(kg1_triples, kg1_elements, kg2_triples, kg2_unlinked, ordered=ordered)
        for ele in kg2_linked:
            ids2[ele] = ids1[linked[ele]]
    else:
        index = 0
        for e1, e2 in train_links:
            assert e1 in kg1_elements and e2 in kg2_elements
            ids1[e1] = index
            ids2[e2] = index
            index += 1
        for ele in kg1_elements:
            if ele not in ids1:
                ids1[ele] = index
                index += 1
        for ele in kg2_elements:
            if ele not in ids2:
                ids2[ele] = index
                index += 1
    assert len(ids1) == len(set(kg1_elements))
    assert len(ids2) == len(set(kg2_elements))
    return ids1, ids2


def uris_list_2ids(uris, ids):
    out = [ids[u] for u in uris]
    assert len(out) == len(set(uris))
    return out


def uris_pair_2ids(uris, ids1, ids2):
    """read.py:104-112: pairs with an unknown side are silently dropped."""
    return [(ids1[u1], ids2[u2]) for u1, u2 in uris if u1 in ids1 and u2 in ids2]


def uris_relation_triple_2ids(uris, ent_ids, rel_ids):
    out = [(ent_ids[h], rel_ids[r], ent_ids[t]) for h, r, t in uris]
    assert len(out) == len(set(uris))
    return out


def uris_attribute_triple_2ids(uris, ent_ids, attr_ids):
    out = [(ent_ids[e], attr_ids[a], v) for e, a, v in uris]
    assert len(out) == len(set(uris))
    return out


def generate_sup_relation_triples_one_link(e1, e2, rt_dict, hr_dict):
    """read.py:136-142: e1's triples with e1 swapped for e2."""
    new = {(e2, r, t) for r, t in rt_dict.get(e1, ())}
    new |= {(h, r, e2) for h, r in hr_dict.get(e1, ())}
    return new


def generate_sup_relation_triples(sup_links, rt_dict1, hr_dict1, rt_dict2, hr_dict2):
    """read.py:145-151."""
    new1, new2 = set(), set()
    for ent1, ent2 in sup_links:
        new1 |= generate_sup_relation_triples_one_link(ent1, ent2, rt_dict1, hr_dict1)
        new2 |= generate_sup_relation_triples_one_link(ent2, ent1, rt_dict2, hr_dict2)
    print("supervised relation triples: {}, {}".format(len(new1), len(new2)))
    return new1, new2


def generate_sup_attribute_triples(sup_links, av_dict1, av_dict2):
    """read.py:162-168."""
    new1, new2 = set(), set()
    for ent1, ent2 in sup_links:
        new1 |= {(ent2, a, v) for a, v in av_dict1.get(ent1, ())}
        new2 |= {(ent1, a, v) for a, v in av_dict2.get(ent2, ())}
    print("supervised attribute triples: {}, {}".format(len(new1), len(new2)))
    return new1, new2


def read_relation_triples(file_path):
    """read.py:222-239: tab-separated (h, r, t) URIs."""
    print("read relation triples:", file_path)
    if file_path is None:
        return set(), set(), set()
    triples, entities, relations = set(), set(), set()
    with open(file_path, 'r', encoding='utf8') as f:
        for line in f:
            params = line.strip('\n').split('\t')
            assert len(params) == 3
            h, r, t = (p.strip() for p in params)
            triples.add((h, r, t))
            entities.add(h)
            entities.add(t)
            relations.add(r)
    return triples, entities, relations


def read_attribute_triples(file_path):
    """read.py:368-391."""
    print("read attribute triples:", file_path)
    if file_path is None or not os.path.exists(file_path):
        return set(), set(), set()
    triples, entities, attributes = set(), set(), set()
    with open(file_path, 'r', encoding='utf8') as f:
        for line in f:
            params = line.strip().strip('\n').split('\t')
            if len(params) < 3:
                continue
            head, attr = params[0].strip(), params[1].strip()
            value = ' '.join(p.strip() for p in params[2:])
            value = value.strip().rstrip('.').strip()
            entities.add(head)
            attributes.add(attr)
            triples.add((head, attr, value))
    return triples, entities, attributes


def read_links(file_path):
    """read.py:242-257."""
    print("read links:", file_path)
    links = []
    with open(file_path, 'r', encoding='utf8') as f:
        for line in f:
            params = line.strip('\n').split('\t')
            assert len(params) == 2
            links.append((params[0].strip(), params[1].strip()))
    return links


def read_dict(file_path):
    ids = {}
    with open(file_path, 'r', encoding='utf8') as f:
        for line in f:
            params = line.strip('\n').split('\t')
            assert len(params) == 2
            ids[params[0]] = int(params[1])
    return ids


def read_pair_ids(file_path):
    pairs = []
    with open(file_path, 'r', encoding='utf8') as f:
        for line in f:
            params = line.strip('\n').split('\t')
            assert len(params) == 2
            pairs.append((int(params[0]), int(params[1])))
    return pairs


def pair2file(file, pairs):
    if pairs is None:
        return
    with open(file, 'w', encoding='utf8') as f:
        for i, j in pairs:
            f.write(str(i) + '\t' + str(j) + '\n')


def dict2file(file, dic):
    if dic is None:
        return
    with open(file, 'w', encoding='utf8') as f:
        for i, j in dic.items():
            f.write(str(i) + '\t' + str(j) + '\n')
    print(file, "saved.")


def line2file(file, lines):
    if lines is None:
        return
    with open(file, 'w', encoding='utf8') as f:
        for line in lines:
            f.write(line + '\n')
    print(file, "saved.")


def save_results(folder, rest_12):
    """read.py:318-322."""
    os.makedirs(folder, exist_ok=True)
    pair2file(folder + 'alignment_results_12', rest_12)
    print("Results saved!")


def embed2file(results_folder, file_name, embedding, kg1_id_dict, kg2_id_dict, seperate=True):
    """read.py:351-366: `uri v0 v1 ...` per line (np.savetxt would change the float repr; the
    reference uses str() of each numpy scalar, kept here)."""
    if embedding is None or kg1_id_dict is None or kg2_id_dict is None:
        return

    def dump(path, dicts):
        with open(path, 'w', encoding='utf8') as f:
            for dic in dicts:
                for uri, index in dic.items():
                    f.write(str(uri) + ' ' + ' '.join(map(str, embedding[index])) + '\n')
    if seperate:
        dump(results_folder + 'kg1_' + file_name, [kg1_id_dict])
        dump(results_folder + 'kg2_' + file_name, [kg2_id_dict])
    else:
        dump(results_folder + 'combined_' + file_name, [kg1_id_dict, kg2_id_dict])


def save_embeddings(folder, kgs, ent_embeds, rel_embeds, attr_embeds, mapping_mat=None, rev_mapping_mat=None):
    """read.py:325-349: same file names, same .npy payload ([rows, dim] C-contiguous fp32)."""
    os.makedirs(folder, exist_ok=True)
    for name, arr in (('ent_embeds', ent_embeds), ('rel_embeds', rel_embeds), ('attr_embeds', attr_embeds),
                      ('mapping_mat', mapping_mat), ('rev_mapping_mat', rev_mapping_mat)):
        if arr is not None:
            np.save(folder + name + '.npy', arr)
    dict2file(folder + 'kg1_ent_ids', kgs.kg1.entities_id_dict)
    dict2file(folder + 'kg2_ent_ids', kgs.kg2.entities_id_dict)
    dict2file(folder + 'kg1_rel_ids', kgs.kg1.relations_id_dict)
    dict2file(folder + 'kg2_rel_ids', kgs.kg2.relations_id_dict)
    dict2file(folder + 'kg1_attr_ids', kgs.kg1.attributes_id_dict)
    dict2file(folder + 'kg2_attr_ids', kgs.kg2.attributes_id_dict)
    embed2file(folder, 'ent_embeds_txt', ent_embeds, kgs.kg1.entities_id_dict, kgs.kg2.entities_id_dict)
    embed2file(folder, 'rel_embeds_txt', rel_embeds, kgs.kg1.relations_id_dict, kgs.kg2.relations_id_dict)
    embed2file(folder, 'attr_embeds_txt', attr_embeds, kgs.kg1.attributes_id_dict, kgs.kg2.attributes_id_dict)
    print("Embeddings saved!")


def generate_sup_attribute_triples_one_link(e1, e2, av_dict):
    """read.py:154-158: e1's attribute triples re-headed to e2."""
    return {(e2, a, v) for a, v in av_dict.get(e1, set())}


def radio_2file(radio, folder):
    """read.py:311-315: <folder><ratio with '_' for '.'>/ (created)."""
    path = folder + str(radio).replace('.', '_')
    os.makedirs(path, exist_ok=True)
    return path + '/'
