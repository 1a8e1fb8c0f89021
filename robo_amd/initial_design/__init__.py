"""Initial designs (host side, O(N D)): the reference's four functions under the reference's module paths
(robo/initial_design/__init__.py:1-4) -- same signatures and draw order, so seeded runs pick the same points."""
from robo_amd.initial_design.init_grid import init_grid  # noqa: F401
from robo_amd.initial_design.init_random_uniform import init_random_uniform  # noqa: F401
from robo_amd.initial_design.init_latin_hypercube_sampling import init_latin_hypercube_sampling  # noqa: F401
from robo_amd.initial_design.init_random_normal import init_random_normal  # noqa: F401
