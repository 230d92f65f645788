"""Primitive-name <-> id registry (drop-in for SPFN/fitter_factory.py:5-30).  The
`create_primitive_from_dict` dispatcher builds host-side numpy containers
(SPFN/primitives.py) and is outside the hot path."""
from . import cone_fitter, cylinder_fitter, plane_fitter, sphere_fitter  # noqa: F401

primitive_name_to_id_dict = {}


def primitive_name_to_id(name):
    return primitive_name_to_id_dict[name]


def get_n_registered_primitives():
    return len(primitive_name_to_id_dict)


def register_primitives(primitive_name_list):
    """Must be called once before everything; ids follow the list order
    (Configs/config_globalSPFN.yml:13-17: sphere, plane, cylinder, cone)."""
    global primitive_name_to_id_dict
    primitive_name_to_id_dict = {name: i for i, name in enumerate(primitive_name_list)}
    print('Registered ' + ','.join(primitive_name_list))


def create_primitive_from_dict(d):
    raise NotImplementedError("host-side primitive containers are outside the MI355X hot path "
                              "(SURVEY.md §2 row 8); use the reference's SPFN.primitives")
