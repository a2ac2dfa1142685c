"""Where the single-colour tables the reference SHIPS differ from the format's rule.

Bit-exact parity needs the shipped behaviour, so tools/gen_bc7_single_color.py applies these
(found by tools/check_tables_vs_reference.py; the reference's table tool passes the same
arguments, MakeTables/Program.cs:355, 405-408):
  * mode 0, p-bits (1,0), index 3 is built with p-bits (1,1) (and is tagged p-bits 3);
  * the four mode-7 tables are built for 7-bit endpoints although mode 7 stores 5 bits.
Key = position in gen_bc7_single_color.SPEC."""
OVERRIDES = {
    8: (0, 4, 2, 1, 1, 3, 7),
    35: (7, 7, 2, 0, 0, 1, 3),
    36: (7, 7, 2, 0, 1, 1, 3),
    37: (7, 7, 2, 1, 0, 1, 3),
    38: (7, 7, 2, 1, 1, 1, 3),
}
