/*
 * ref_task_harness.cpp -- TEST INFRASTRUCTURE: the reference's file-sink program, compiled from the reference's own text.
 *
 * The reference as a whole needs UHD and Boost, which this image lacks; a build against stand-ins for them is not allowed.
 * But only FOUR places of it name them -- include/structures.h:2 (#include <uhd/...>), :18-19 (two members of tx_t, the USRP
 * sender's state), include/galileo-sdr.h:12-15 (four Boost includes nothing on the file path uses) and src/main.cpp:55-127
 * (tx_task, the USRP sender) plus src/usrp.cpp -- and `-U 1` (file sink, src/main.cpp:308-311) never reaches them.
 * oracle/Makefile (target _ref/ref_task, only when /root/reference is present) therefore cuts, at build time and from where
 * they lie, into a git-ignored scratch directory oracle/_ref/gen/ that is deleted again after the compile:
 *   gen/include/structures.h    include/structures.h  WITHOUT lines 2, 18, 19
 *   gen/include/galileo-sdr.h   include/galileo-sdr.h WITHOUT lines 12-15
 *   gen/src/<name>.cpp          src/{galileo-sdr,geodesy,gnss-time,iono,rinex,inav-msg,channel,gal-sig,datatypes,debug,fifo}.cpp,
 *                               every line
 *   gen/main_part.inc           src/main.cpp:1-54,128-409 (everything but tx_task)
 * Every other header is the reference's own, found through -I/root/reference/include (constants.h, socket.h), or the
 * system's (ncurses.h and libncurses ARE in this image).  Lines are only ever OMITTED, nothing is rewritten and no header or
 * library is stood in for.  The standard headers that <uhd/...> and Boost pull in for the reference transitively come in by
 * -include on the command line (vector, map, string, queue, pthread.h, math.h, sys/time.h).  Reference flags:
 * -std=c++11 -g -DDEBUG, no -O (CMakeLists.txt:18-22).
 *
 * This file supplies the two names main() still mentions and `-U 1` never calls.  Nothing of the reference is committed.
 *
 * Use (tools/ref_task_goldens.py, tests/test_ref_task.py): `TERM=xterm oracle/_ref/ref_task -e <rinex> -l .. -t .. -d .. -U 1 -b 1
 * -o <file>`: the md5 of <file> is the reference's answer.  (After the file is complete and closed the program ends in
 * std::terminate: galileo_task leaves the joinable std::thread th_loc of src/galileo-sdr.cpp:185 behind -- the reference's
 * own exit path; the exit status is 134.)
 */
#include "galileo-sdr.h" /* gen/include: the reference's, four include lines shorter */
#include <cstdio>
#include <cstdlib>

/* src/main.cpp:55-127 (UHD sender) and src/usrp.cpp:init_usrp: not built; main() calls them only without -U 1 */
void *tx_task(void *)
{
    fprintf(stderr, "ref_task: the USRP path is not built; run with -U 1\n");
    abort();
}
void init_usrp(usrp_conf_t, sim_t *)
{
    fprintf(stderr, "ref_task: the USRP path is not built; run with -U 1\n");
    abort();
}

#include "_ref/gen/main_part.inc" /* src/main.cpp:1-54,128-409, verbatim */
