"""In-memory KG: the attribute surface of openea/modules/load/kg.py:10-141 (every `<x>_set`, `<x>_list`,
`<x>_num` and `*_dict` a model or a loader of the reference reads), built from two generic helpers.

Every list is in SORTED order.  The reference takes `list(set)` (kg.py:58,63,...), an arbitrary order that
changes with PYTHONHASHSEED for URI strings; data-parallel ranks (one process each) must lay out identical
batches and candidate lists, so the order is fixed here."""
from collections import defaultdict


def parse_triples(triples):
    """The three column sets of a triple collection (kg.py:1-7)."""
    columns = (set(), set(), set())
    for triple in triples:
        for column, item in zip(columns, triple):
            column.add(item)
    return columns


def _grouped(triples, key, cols):
    """{triple[key]: {triple[cols]}} (cols an index) or {triple[key]: {(triple[a], triple[b])}} (cols = (a, b)) as a plain
    dict of sets."""
    groups = defaultdict(set)
    if isinstance(cols, int):
        for triple in triples:
            groups[triple[key]].add(triple[cols])
    else:
        a, b = cols
        for triple in triples:
            groups[triple[key]].add((triple[a], triple[b]))
    return dict(groups)


class KG:
    _STATS = (("entities", "entities"), ("relations", "relations"), ("attributes", "attributes"),
              ("relation triples", "relation_triples"), ("attribute triples", "attribute_triples"),
              ("local relation triples", "local_relation_triples"),
              ("local attribute triples", "local_attribute_triples"))

    def __init__(self, relation_triples, attribute_triples, verbose=True):
        for name in ("entities", "relations", "attributes"):
            setattr(self, name + "_id_dict", None)
        for kind in ("relation", "attribute"):
            setattr(self, "sup_%s_triples_set" % kind, None)
            setattr(self, "sup_%s_triples_list" % kind, None)
        self.set_relations(relation_triples)
        self.set_attributes(attribute_triples)
        if verbose:
            print("\nKG statistics:")
            for label, name in self._STATS:
                print("Number of %s:" % label, getattr(self, name + "_num"))
            print()

    def _publish(self, name, items, aliases=()):
        """<name>_set / _list / _num (and the same objects under each alias)."""
        as_set = items if isinstance(items, set) else set(items)
        as_list = sorted(as_set)
        for prefix in (name,) + tuple(aliases):
            setattr(self, prefix + "_set", as_set)
            setattr(self, prefix + "_list", as_list)
            setattr(self, prefix + "_num", len(as_list))

    def set_relations(self, relation_triples):
        """kg.py:56-72: the local triples ARE the training triples until sup triples are added."""
        self._publish("relation_triples", set(relation_triples), aliases=("local_relation_triples",))
        heads, relations, tails = parse_triples(self.relation_triples_set)
        self._publish("entities", heads | tails)
        self._publish("relations", relations)
        self._lazy = {}                      # the dictionaries below are built at first use (see _lazy_dict)

    def set_attributes(self, attribute_triples):
        """kg.py:74-93 (entities that only occur in attribute triples join the entity set)."""
        self._publish("attribute_triples", set(attribute_triples), aliases=("local_attribute_triples",))
        subjects, attributes, _ = parse_triples(self.attribute_triples_set)
        self._publish("attributes", attributes)
        self.entities_set |= subjects
        self._publish("entities", self.entities_set)
        for name in ("av_dict", "entity_attributes_dict"):
            self._lazy.pop(name, None)

    # The reference builds these five dictionaries in the constructor (kg.py:95-130) -- also for the URI-keyed KG objects
    # that read_kgs_from_folder only needs for their triple / element sets.  They are pure functions of the LOCAL triple
    # lists (which add_sup_* never changes), so building them at first access gives the same objects and saves the loader
    # a third of its time at the 100K shape.
    def _lazy_dict(self, name, build):
        if name not in self._lazy:
            self._lazy[name] = build()
        return self._lazy[name]

    @property
    def rt_dict(self):
        """kg.py:95-105: rt_dict[h] = {(r, t)}."""
        return self._lazy_dict("rt_dict", lambda: _grouped(self.local_relation_triples_list, 0, (1, 2)))

    @property
    def hr_dict(self):
        """kg.py:95-105: hr_dict[t] = {(h, r)}."""
        return self._lazy_dict("hr_dict", lambda: _grouped(self.local_relation_triples_list, 2, (0, 1)))

    @property
    def av_dict(self):
        """kg.py:107-114: av_dict[e] = {(a, v)}."""
        return self._lazy_dict("av_dict", lambda: _grouped(self.local_attribute_triples_list, 0, (1, 2)))

    @property
    def entity_relations_dict(self):
        """kg.py:116-122."""
        return self._lazy_dict("entity_relations_dict", lambda: _grouped(self.local_relation_triples_list, 0, 1))

    @property
    def entity_attributes_dict(self):
        """kg.py:124-130."""
        return self._lazy_dict("entity_attributes_dict", lambda: _grouped(self.local_attribute_triples_list, 0, 1))

    def generate_relation_triple_dict(self):
        self._lazy.pop("rt_dict", None)
        self._lazy.pop("hr_dict", None)
        return self.rt_dict, self.hr_dict

    def generate_attribute_triple_dict(self):
        self._lazy.pop("av_dict", None)
        return self.av_dict

    def parse_relations(self):
        self._lazy.pop("entity_relations_dict", None)
        return self.entity_relations_dict

    def parse_attributes(self):
        self._lazy.pop("entity_attributes_dict", None)
        return self.entity_attributes_dict

    def set_id_dict(self, entities_id_dict, relations_id_dict, attributes_id_dict):
        self.entities_id_dict, self.relations_id_dict, self.attributes_id_dict = \
            entities_id_dict, relations_id_dict, attributes_id_dict

    def _add_sup(self, kind, sup_triples):
        """The seed-swapped triples join the training triples IN PLACE, as kg.py:136-141 does: the
        local_*_set is the same set object and grows with it (the aliasing of kg.py:59/77, kept because
        the golden loader statistics come from the reference), while local_*_list/_num and the dicts built
        from them keep the pre-swap triples."""
        sup = set(sup_triples)
        setattr(self, "sup_%s_triples_set" % kind, sup)
        setattr(self, "sup_%s_triples_list" % kind, sorted(sup))
        training = getattr(self, "%s_triples_set" % kind)
        training |= sup
        self._publish("%s_triples" % kind, training)

    def add_sup_relation_triples(self, sup_triples):
        """kg.py:136-141."""
        self._add_sup("relation", sup_triples)

    def add_sup_attribute_triples(self, sup_triples):
        self._add_sup("attribute", sup_triples)
