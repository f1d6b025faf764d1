// gd_ab_sphere -- sphere-confined A/B copolymer driver (4-sim-ab/sphere/src); see gd_ab_driver.hpp
#include "gd_ab_driver.hpp"
int main(int argc, char **argv) { return gd_ab::main_ab(gd_ab::geometry::sphere, argc, argv); }
