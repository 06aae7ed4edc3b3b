"""Pair of KGs in id space (mirror of openea/modules/load/kgs.py:5-99)."""
import os

from .kg import KG
from .read import (generate_mapping_id, generate_sharing_id, generate_sup_attribute_triples,
                   generate_sup_relation_triples, read_attribute_triples, read_links,
                   read_relation_triples, uris_attribute_triple_2ids, uris_pair_2ids,
                   uris_relation_triple_2ids)


class KGs:
    def __init__(self, kg1: KG, kg2: KG, train_links, test_links, valid_links=None, mode='mapping', ordered=True,
                 verbose=True):
        if mode == "sharing":       # kgs.py:7-13
            gen = generate_sharing_id
            ent_ids1, ent_ids2 = gen(train_links, kg1.relation_triples_set, kg1.entities_set,
                                     kg2.relation_triples_set, kg2.entities_set, ordered=ordered)
            rel_ids1, rel_ids2 = gen([], kg1.relation_triples_set, kg1.relations_set,
                                     kg2.relation_triples_set, kg2.relations_set, ordered=ordered)
            attr_ids1, attr_ids2 = gen([], kg1.attribute_triples_set, kg1.attributes_set,
                                       kg2.attribute_triples_set, kg2.attributes_set, ordered=ordered)
        else:                       # kgs.py:14-20
            ent_ids1, ent_ids2 = generate_mapping_id(kg1.relation_triples_set, kg1.entities_set,
                                                     kg2.relation_triples_set, kg2.entities_set, ordered=ordered)
            rel_ids1, rel_ids2 = generate_mapping_id(kg1.relation_triples_set, kg1.relations_set,
                                                     kg2.relation_triples_set, kg2.relations_set, ordered=ordered)
            attr_ids1, attr_ids2 = generate_mapping_id(kg1.attribute_triples_set, kg1.attributes_set,
                                                       kg2.attribute_triples_set, kg2.attributes_set, ordered=ordered)
        id_rel1 = uris_relation_triple_2ids(kg1.relation_triples_set, ent_ids1, rel_ids1)
        id_rel2 = uris_relation_triple_2ids(kg2.relation_triples_set, ent_ids2, rel_ids2)
        id_attr1 = uris_attribute_triple_2ids(kg1.attribute_triples_set, ent_ids1, attr_ids1)
        id_attr2 = uris_attribute_triple_2ids(kg2.attribute_triples_set, ent_ids2, attr_ids2)

        self.uri_kg1, self.uri_kg2 = kg1, kg2
        kg1 = KG(id_rel1, id_attr1, verbose=verbose)
        kg2 = KG(id_rel2, id_attr2, verbose=verbose)
        kg1.set_id_dict(ent_ids1, rel_ids1, attr_ids1)
        kg2.set_id_dict(ent_ids2, rel_ids2, attr_ids2)

        self.uri_train_links, self.uri_test_links = train_links, test_links
        self.train_links = uris_pair_2ids(train_links, ent_ids1, ent_ids2)
        self.test_links = uris_pair_2ids(test_links, ent_ids1, ent_ids2)
        self.train_entities1 = [l[0] for l in self.train_links]
        self.train_entities2 = [l[1] for l in self.train_links]
        self.test_entities1 = [l[0] for l in self.test_links]
        self.test_entities2 = [l[1] for l in self.test_links]

        if mode == 'swapping':      # kgs.py:45-54
            sup1, sup2 = generate_sup_relation_triples(self.train_links, kg1.rt_dict, kg1.hr_dict,
                                                       kg2.rt_dict, kg2.hr_dict)
            kg1.add_sup_relation_triples(sup1)
            kg2.add_sup_relation_triples(sup2)
            sup1, sup2 = generate_sup_attribute_triples(self.train_links, kg1.av_dict, kg2.av_dict)
            kg1.add_sup_attribute_triples(sup1)
            kg2.add_sup_attribute_triples(sup2)

        self.kg1, self.kg2 = kg1, kg2
        self.valid_links, self.valid_entities1, self.valid_entities2 = [], [], []
        if valid_links is not None:
            self.uri_valid_links = valid_links
            self.valid_links = uris_pair_2ids(valid_links, ent_ids1, ent_ids2)
            self.valid_entities1 = [l[0] for l in self.valid_links]
            self.valid_entities2 = [l[1] for l in self.valid_links]

        self.useful_entities_list1 = self.kg1.entities_list       # kgs.py:71-72
        self.useful_entities_list2 = self.kg2.entities_list
        self.entities_num = len(self.kg1.entities_set | self.kg2.entities_set)
        self.relations_num = len(self.kg1.relations_set | self.kg2.relations_set)
        self.attributes_num = len(self.kg1.attributes_set | self.kg2.attributes_set)


def remove_unlinked_triples(triples, links):
    """kgs.py:211-222."""
    linked = set()
    for i, j in links:
        linked.add(i)
        linked.add(j)
    return {(h, r, t) for h, r, t in triples if h in linked and t in linked}


def _read_standard_folder(folder, division, mode, ordered, remove_unlinked, reverse):
    """rel_triples_{1,2}, attr_triples_{1,2}, <division>{train,valid,test}_links; reverse=True swaps the roles of the two
    KGs and the direction of every link (kgs.py:102-123)."""
    first, second = ('2', '1') if reverse else ('1', '2')
    rels = [read_relation_triples(folder + 'rel_triples_' + side)[0] for side in (first, second)]
    attrs = [read_attribute_triples(folder + 'attr_triples_' + side)[0] for side in (first, second)]
    links = {}
    for part in ('train', 'valid', 'test'):
        pairs = read_links(folder + division + part + '_links')
        links[part] = [(j, i) for i, j in pairs] if reverse else pairs
    if remove_unlinked:
        every = links['train'] + links['valid'] + links['test']
        rels = [remove_unlinked_triples(r, every) for r in rels]
    return KGs(KG(rels[0], attrs[0]), KG(rels[1], attrs[1]), links['train'], links['test'], valid_links=links['valid'],
               mode=mode, ordered=ordered)


def read_kgs_from_folder(training_data_folder, division, mode, ordered, remove_unlinked=False):
    """kgs.py:79-99 (DBP15K / DWY100K folders go to read_kgs_from_dbp_dwy, :80-81)."""
    lowered = training_data_folder.lower()
    if 'dbp15k' in lowered or 'dwy100k' in lowered:
        return read_kgs_from_dbp_dwy(training_data_folder, division, mode, ordered, remove_unlinked=remove_unlinked)
    return _read_standard_folder(training_data_folder, division, mode, ordered, remove_unlinked, reverse=False)


def read_reversed_kgs_from_folder(training_data_folder, division, mode, ordered, remove_unlinked=False):
    """kgs.py:102-123: KG2 takes the role of KG1 and every link is turned round (run/main_from_args_reversed.py)."""
    return _read_standard_folder(training_data_folder, division, mode, ordered, remove_unlinked, reverse=True)


def remove_no_triples_link(kg1_relation_triples, kg2_relation_triples, train_links, test_links):
    """kgs.py:172-189: links whose two entities both still occur in a relation triple."""
    ents1 = {e for h, _, t in kg1_relation_triples for e in (h, t)}
    ents2 = {e for h, _, t in kg2_relation_triples for e in (h, t)}
    print("before removing links with no triples:", len(train_links), len(test_links))
    kept = [list({(i, j) for i, j in part if i in ents1 and j in ents2}) for part in (train_links, test_links)]
    print("after removing links with no triples:", len(kept[0]), len(kept[1]))
    return kept[0], kept[1]


def read_kgs_from_dbp_dwy(folder, division, mode, ordered, remove_unlinked=False):
    """kgs.py:134-169: the DBP15K / DWY100K layout -- triples_{1,2}, sup_pairs | sup_ent_ids, ref_pairs | ref_ent_ids under
    <folder><division>, no attribute triples, no validation links; remove_unlinked alternates the two filters until
    neither removes anything."""
    folder = folder + division
    rel1 = read_relation_triples(folder + 'triples_1')[0]
    rel2 = read_relation_triples(folder + 'triples_2')[0]
    train_links = read_links(folder + ('sup_pairs' if os.path.exists(folder + 'sup_pairs') else 'sup_ent_ids'))
    test_links = read_links(folder + ('ref_pairs' if os.path.exists(folder + 'ref_pairs') else 'ref_ent_ids'))
    print()
    while remove_unlinked:
        rel1 = remove_unlinked_triples(rel1, train_links + test_links)
        rel2 = remove_unlinked_triples(rel2, train_links + test_links)
        before = (len(rel1), len(rel2))
        train_links, test_links = remove_no_triples_link(rel1, rel2, train_links, test_links)
        rel1 = remove_unlinked_triples(rel1, train_links + test_links)
        rel2 = remove_unlinked_triples(rel2, train_links + test_links)
        if before == (len(rel1), len(rel2)):
            break
    return KGs(KG(rel1, list()), KG(rel2, list()), train_links, test_links, mode=mode, ordered=ordered)


def read_kgs_from_files(kg1_relation_triples, kg2_relation_triples, kg1_attribute_triples, kg2_attribute_triples,
                        train_links, valid_links, test_links, mode):
    """kgs.py:128-133."""
    kg1 = KG(kg1_relation_triples, kg1_attribute_triples)
    kg2 = KG(kg2_relation_triples, kg2_attribute_triples)
    return KGs(kg1, kg2, train_links, test_links, valid_links=valid_links, mode=mode)
