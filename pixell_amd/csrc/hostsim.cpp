// thread-local state and the fiber scheduler of the TEST-ONLY host simulator (see hostsim.hpp)
#ifdef PXS_HOST_SIM
#include "hostsim.hpp"
#include <sys/mman.h>
#include <cstdio>
#if !defined(__x86_64__)
#error "the host simulator's context switch is written for x86-64"
#endif
// save the callee-saved registers and the stack pointer of the running context in *save_sp, continue the context whose stack pointer is load_sp
// (swapcontext would do, at two signal-mask system calls per switch)
extern "C" void pxsim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl pxsim_switch
.type pxsim_switch,@function
pxsim_switch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq %rsi, %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
.size pxsim_switch,.-pxsim_switch
)");
namespace pxsim {
thread_local uint3_ t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_ctx;

bool fiber_mode() { static const bool on = [] { const char* e = getenv("PXS_SIM_THREADS"); return !(e && atoi(e) != 0); }(); return on; }
int sim_workers() { static const int n = [] { const char* e = getenv("PXS_SIM_WORKERS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency())); }(); return n; }

// the lanes of the block this OS thread is running: contexts on private stacks, scheduled round robin; a lane gives up the thread only inside
// fiber_yield() (a barrier or wave rendezvous it cannot pass yet) and when it ends
struct Fiber { void* sp; uint3_ tid; bool done; };
struct FiberSet {
	std::vector<Fiber> lanes; char* stacks = nullptr; size_t stack_bytes = 0; int cur = -1, alive = 0; void* main_sp = nullptr;
	const std::function<void()>* body = nullptr;
	~FiberSet() { if (stacks) munmap(stacks, stack_bytes); }
};
static thread_local FiberSet* t_fs = nullptr;
static constexpr size_t FIBER_STACK = 256*1024;

static void fiber_switch_from(FiberSet* fs, int from) {
	// the next lane after `from` that has not ended; the scheduler context when none is left
	const int n = (int)fs->lanes.size();
	int nxt = -1;
	for (int k = 1; k <= n; k++) { const int c = (from + k) % n; if (!fs->lanes[c].done) { nxt = c; break; } }
	void* dummy;
	if (nxt < 0) { fs->cur = -1; pxsim_switch(&dummy, fs->main_sp); }
	if (nxt == from) { fprintf(stderr, "hostsim: lane %d waits at a rendezvous no other lane can reach (deadlock)\n", from); abort(); }
	fs->cur = nxt;
	pxsim_switch(fs->lanes[from].done ? &dummy : &fs->lanes[from].sp, fs->lanes[nxt].sp);
	t_threadIdx = fs->lanes[from].tid;      // (resumed)
}
void fiber_yield() { FiberSet* fs = t_fs; fiber_switch_from(fs, fs->cur); }
static void fiber_entry() {
	FiberSet* fs = t_fs; const int me = fs->cur;
	t_threadIdx = fs->lanes[me].tid;
	(*fs->body)();
	fs->lanes[me].done = true; fs->alive--;
	fiber_switch_from(fs, me);
	abort();
}
void run_block_fibers(int nt, dim3 block, const std::function<void()>& body) {
	static thread_local FiberSet fsl;
	FiberSet* fs = &fsl; t_fs = fs;
	if ((int)fs->lanes.size() != nt) {
		if (fs->stacks) munmap(fs->stacks, fs->stack_bytes);
		fs->lanes.assign(nt, Fiber()); fs->stack_bytes = (size_t)nt*FIBER_STACK;
		fs->stacks = (char*)mmap(nullptr, fs->stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (fs->stacks == (char*)MAP_FAILED) { perror("hostsim: mmap of the lane stacks"); abort(); }
	}
	fs->body = &body; fs->alive = nt;
	for (int t = 0; t < nt; t++) {
		Fiber& f = fs->lanes[t];
		f.tid = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x*block.y))}; f.done = false;
		// a fresh context: six zeroed callee-saved registers, then the address pxsim_switch returns to, on a 16-byte aligned slot
		void** top = reinterpret_cast<void**>(fs->stacks + (size_t)(t + 1)*FIBER_STACK);
		top[-1] = nullptr; top[-2] = reinterpret_cast<void*>(&fiber_entry);
		for (int k = 3; k <= 8; k++) top[-k] = nullptr;
		f.sp = top - 8;
	}
	fs->cur = 0;
	pxsim_switch(&fs->main_sp, fs->lanes[0].sp);      // comes back when the last lane has ended
}
}
#endif
