"""In-memory KG (mirror of openea/modules/load/kg.py: same attribute names, kg.py:10-141)."""


def _ordered(s):
    """list of a set in SORTED order.  The reference takes `list(set)` (kg.py:58,63,...), i.e. an arbitrary
    order that changes with PYTHONHASHSEED for URI strings; data-parallel ranks (one process each) must
    lay out identical batches and candidate lists, so the order is fixed here."""
    return sorted(s)


def parse_triples(triples):
    subjects, predicates, objects = set(), set(), set()
    for s, p, o in triples:
        subjects.add(s)
        predicates.add(p)
        objects.add(o)
    return subjects, predicates, objects


class KG:
    def __init__(self, relation_triples, attribute_triples, verbose=True):
        self.entities_id_dict = None
        self.relations_id_dict = None
        self.attributes_id_dict = None
        self.sup_relation_triples_set, self.sup_relation_triples_list = None, None
        self.sup_attribute_triples_set, self.sup_attribute_triples_list = None, None
        self.set_relations(relation_triples)
        self.set_attributes(attribute_triples)
        if verbose:
            print()
            print("KG statistics:")
            print("Number of entities:", self.entities_num)
            print("Number of relations:", self.relations_num)
            print("Number of attributes:", self.attributes_num)
            print("Number of relation triples:", self.relation_triples_num)
            print("Number of attribute triples:", self.attribute_triples_num)
            print("Number of local relation triples:", self.local_relation_triples_num)
            print("Number of local attribute triples:", self.local_attribute_triples_num)
            print()

    def set_relations(self, relation_triples):
        """kg.py:56-72."""
        self.relation_triples_set = set(relation_triples)
        self.relation_triples_list = _ordered(self.relation_triples_set)
        self.local_relation_triples_set = self.relation_triples_set
        self.local_relation_triples_list = self.relation_triples_list
        heads, relations, tails = parse_triples(self.relation_triples_set)
        self.entities_set = heads | tails
        self.relations_set = relations
        self.entities_list = _ordered(self.entities_set)
        self.relations_list = _ordered(self.relations_set)
        self.entities_num = len(self.entities_set)
        self.relations_num = len(self.relations_set)
        self.relation_triples_num = len(self.relation_triples_set)
        self.local_relation_triples_num = len(self.local_relation_triples_set)
        self.generate_relation_triple_dict()
        self.parse_relations()

    def set_attributes(self, attribute_triples):
        """kg.py:74-93 (entities that only occur in attribute triples join the entity set)."""
        self.attribute_triples_set = set(attribute_triples)
        self.attribute_triples_list = _ordered(self.attribute_triples_set)
        self.local_attribute_triples_set = self.attribute_triples_set
        self.local_attribute_triples_list = self.attribute_triples_list
        entities, attributes, _ = parse_triples(self.attribute_triples_set)
        self.attributes_set = attributes
        self.attributes_list = _ordered(self.attributes_set)
        self.attributes_num = len(self.attributes_set)
        self.entities_set |= entities
        self.entities_list = _ordered(self.entities_set)
        self.entities_num = len(self.entities_set)
        self.attribute_triples_num = len(self.attribute_triples_set)
        self.local_attribute_triples_num = len(self.local_attribute_triples_set)
        self.generate_attribute_triple_dict()
        self.parse_attributes()

    def generate_relation_triple_dict(self):
        """kg.py:95-105: rt_dict[h] = {(r,t)}, hr_dict[t] = {(h,r)}."""
        self.rt_dict, self.hr_dict = dict(), dict()
        for h, r, t in self.local_relation_triples_list:
            self.rt_dict.setdefault(h, set()).add((r, t))
            self.hr_dict.setdefault(t, set()).add((h, r))

    def generate_attribute_triple_dict(self):
        self.av_dict = dict()
        for h, a, v in self.local_attribute_triples_list:
            self.av_dict.setdefault(h, set()).add((a, v))

    def parse_relations(self):
        self.entity_relations_dict = dict()
        for ent, rel, _ in self.local_relation_triples_set:
            self.entity_relations_dict.setdefault(ent, set()).add(rel)

    def parse_attributes(self):
        self.entity_attributes_dict = dict()
        for ent, attr, _ in self.local_attribute_triples_set:
            self.entity_attributes_dict.setdefault(ent, set()).add(attr)

    def set_id_dict(self, entities_id_dict, relations_id_dict, attributes_id_dict):
        self.entities_id_dict = entities_id_dict
        self.relations_id_dict = relations_id_dict
        self.attributes_id_dict = attributes_id_dict

    def add_sup_relation_triples(self, sup_triples):
        """kg.py:136-141 (seed-swapped triples join the training triples, not rt/hr_dict)."""
        self.sup_relation_triples_set = set(sup_triples)
        self.sup_relation_triples_list = _ordered(self.sup_relation_triples_set)
        self.relation_triples_set |= sup_triples
        self.relation_triples_list = _ordered(self.relation_triples_set)
        self.relation_triples_num = len(self.relation_triples_list)

    def add_sup_attribute_triples(self, sup_triples):
        self.sup_attribute_triples_set = set(sup_triples)
        self.sup_attribute_triples_list = _ordered(self.sup_attribute_triples_set)
        self.attribute_triples_set |= sup_triples
        self.attribute_triples_list = _ordered(self.attribute_triples_set)
        self.attribute_triples_num = len(self.attribute_triples_list)
