from .ocean_grid_generator import build_parser, main

main(**vars(build_parser().parse_args()))
