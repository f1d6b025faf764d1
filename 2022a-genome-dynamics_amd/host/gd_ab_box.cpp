// gd_ab_box -- periodic-box A/B copolymer driver (4-sim-ab/box/src/simulation); see gd_ab_driver.hpp
#include "gd_ab_driver.hpp"
int main(int argc, char **argv) { return gd_ab::main_ab(gd_ab::geometry::box, argc, argv); }
