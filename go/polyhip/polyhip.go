// Package polyhip is the cgo binding of libpolyhip.so (include/polyhip.h), the MI355X
// implementation of poly's search hot path.  UNCOMPILED in the authoring image (no Go toolchain).
//
// Conventions: every function packs Go strings into one contiguous byte buffer + offsets (the
// library never keeps a Go pointer after returning, so the cgo pointer rules hold), calls the
// host-pointer flavour of the entry point, and converts a negative status into an error whose
// text is polyhip_last_error().  POLYHIP_ERR_PANIC is turned into a Go panic with the same
// message the reference would panic with.
package polyhip

/*
#cgo LDFLAGS: -lpolyhip
#include <stdlib.h>
#include "polyhip.h"
*/
import "C"

import (
	"errors"
	"runtime"
	"unsafe"
)

// Pack concatenates sequences and returns the byte buffer and the n+1 offsets.
func Pack(seqs []string) ([]byte, []uint64) {
	offs := make([]uint64, len(seqs)+1)
	total := 0
	for i, s := range seqs {
		total += len(s)
		offs[i+1] = uint64(total)
	}
	buf := make([]byte, 0, total+1)
	for _, s := range seqs {
		buf = append(buf, s...)
	}
	if len(buf) == 0 {
		buf = append(buf, 0) // a valid pointer for cgo even when every sequence is empty
	}
	return buf, offs
}

func check(rc C.int) error {
	if rc == C.POLYHIP_OK {
		return nil
	}
	msg := C.GoString(C.polyhip_last_error())
	if rc == C.POLYHIP_ERR_PANIC {
		panic(msg)
	}
	return errors.New(msg)
}

// call pins the calling goroutine to its OS thread for the duration: the current HIP device and
// polyhip_last_error() are per-thread state.
func call(f func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	return check(f())
}

// ---- one host call over the node's GPUs (include/polyhip.h: polyhip_set_devices) ----
//
// The Go API of the drop-in has ONE call per batch (mash.SketchBatch, align.SmithWatermanBatch, primers.SantaLuciaScan,
// seqhash.HashBatch ...), so the node's GPUs are reached from inside that call: with a device list every host-pointer
// entry point below shards its batch over the list (one worker thread per entry, each on its own device and PCIe
// link) and writes the results straight into the Go slices it was handed.  Nothing else in this package changes:
// no device pointers, no allocator, no process per GPU.

// SetDevices installs the process-wide device list; an empty list returns to the calling thread's current device.
// An id may repeat ([]int{0, 0, 0}: the shards share that GPU -- the fan-out's test mode on a one-GPU box).
func SetDevices(ids []int) error {
	cids := make([]C.int, len(ids)+1)
	for i, d := range ids {
		cids[i] = C.int(d)
	}
	return call(func() C.int { return C.polyhip_set_devices((*C.int)(unsafe.Pointer(&cids[0])), C.int(len(ids))) })
}

// Devices returns the current device list (the POLYHIP_DEVICES environment variable seeds it).
func Devices() []int {
	var cids [64]C.int
	n := int(C.polyhip_get_devices((*C.int)(unsafe.Pointer(&cids[0])), C.int(64)))
	if n > 64 {
		n = 64
	}
	out := make([]int, n)
	for i := range out {
		out[i] = int(cids[i])
	}
	return out
}

// Init is polyhip_init: devices 0..n-1 (n <= 0: every visible device).  Shutdown clears the list.
func Init(nDevices int) error { return call(func() C.int { return C.polyhip_init(C.int(nDevices)) }) }
func Shutdown() error         { return call(func() C.int { return C.polyhip_shutdown() }) }

// IsASCII reports whether every byte is below 0x80.  The C ABI refuses other input wherever the reference would
// case-fold or map it through string(byte) (Go treats such bytes as UTF-8 there: strings.ToUpper turns an invalid byte
// into the three bytes of U+FFFD, primers.go:71,109); the overlays route such sequences to the reference's own bodies
// (kept in the fork as *CPU functions) instead, so the drop-in returns exactly what the reference returns for them.
func IsASCII(s string) bool {
	for i := 0; i < len(s); i++ {
		if s[i] >= 0x80 {
			return false
		}
	}
	return true
}

// MashSketchDistanceMatrix: BASELINE configs[2] in one call -- reads in, sketches (n*s, in/out like MashSketchBatch; nil:
// not wanted) and the all-vs-all matrix out (counts n*n and/or dist n*n; either may be nil).  On a device list the
// reads shard and the devices build ONE index of all sketches together -- index items exchanged by value range over peer
// copies, the sketches themselves never gathered -- then each joins the rows it sketched (a set with a read shorter than
// k + s windows gathers the sketches instead: the reference's merge reads them raw).
func MashSketchDistanceMatrix(seqs []byte, offs []uint64, k, s int, sketches []uint32, counts []uint16, dist []float64) error {
	var ps *C.uint32_t
	var pc *C.uint16_t
	var pd *C.double
	if sketches != nil {
		ps = (*C.uint32_t)(unsafe.Pointer(&sketches[0]))
	}
	if counts != nil {
		pc = (*C.uint16_t)(unsafe.Pointer(&counts[0]))
	}
	if dist != nil {
		pd = (*C.double)(unsafe.Pointer(&dist[0]))
	}
	return call(func() C.int {
		return C.polyhip_mash_sketch_distance_matrix((*C.uint8_t)(unsafe.Pointer(&seqs[0])), (*C.uint64_t)(unsafe.Pointer(&offs[0])),
			C.uint64_t(len(offs)-1), C.uint32_t(k), C.uint32_t(s), ps, pc, pd)
	})
}

// MatrixInfo is polyhip_matrix_info: what the calling OS thread's last MashSketchDistanceMatrix did -- which path, over how
// many devices, its device-to-device copies by transport (Peer: xGMI with peer access on; Staged: no peer access, bounced
// through host memory by the runtime; Local: both ends on one device) and the wall time of its rounds.
type MatrixInfo struct {
	Path, Devices                         int
	PeerCopies, StagedCopies, LocalCopies int
	BytesPeer, BytesStaged, BytesLocal    uint64
	MsSketch, MsIndex, MsJoin             float64
}

// LastMatrixInfo must run on the OS thread that made the call (runtime.LockOSThread around both).
func LastMatrixInfo() (MatrixInfo, error) {
	var ci C.polyhip_matrix_info
	err := call(func() C.int { return C.polyhip_mash_sketch_distance_matrix_last_info((*C.polyhip_matrix_info)(unsafe.Pointer(&ci))) })
	return MatrixInfo{Path: int(ci.path), Devices: int(ci.devices), PeerCopies: int(ci.peer_copies), StagedCopies: int(ci.staged_copies),
		LocalCopies: int(ci.local_copies), BytesPeer: uint64(ci.bytes_peer), BytesStaged: uint64(ci.bytes_staged),
		BytesLocal: uint64(ci.bytes_local), MsSketch: float64(ci.ms_sketch), MsIndex: float64(ci.ms_index), MsJoin: float64(ci.ms_join)}, err
}

// MashSketchBatch: out is n*s uint32, in/out (prior Sketches), see polyhip_mash_sketch_batch.
func MashSketchBatch(seqs []byte, offs []uint64, k, s int, out []uint32) error {
	return call(func() C.int {
		return C.polyhip_mash_sketch_batch((*C.uint8_t)(unsafe.Pointer(&seqs[0])), (*C.uint64_t)(unsafe.Pointer(&offs[0])),
			C.uint64_t(len(offs)-1), C.uint32_t(k), C.uint32_t(s), (*C.uint32_t)(unsafe.Pointer(&out[0])))
	})
}

// MashDistanceMatrix: X is nx*sx, Y is ny*sy; counts (nx*ny) and dist (nx*ny) may be nil.
func MashDistanceMatrix(X []uint32, nx, sx int, Y []uint32, ny, sy int, counts []uint16, dist []float64) error {
	var pc *C.uint16_t
	var pd *C.double
	if counts != nil {
		pc = (*C.uint16_t)(unsafe.Pointer(&counts[0]))
	}
	if dist != nil {
		pd = (*C.double)(unsafe.Pointer(&dist[0]))
	}
	return call(func() C.int {
		return C.polyhip_mash_distance_matrix((*C.uint32_t)(unsafe.Pointer(&X[0])), C.uint64_t(nx), C.uint32_t(sx),
			(*C.uint32_t)(unsafe.Pointer(&Y[0])), C.uint64_t(ny), C.uint32_t(sy), pc, pd)
	})
}

// Scoring wraps polyhip_scoring; Close releases the device tables.
type Scoring struct{ h *C.polyhip_scoring }

func NewScoring(lut *[65536]int32, validA, validB *[256]uint8, gap int) (*Scoring, error) {
	s := &Scoring{}
	err := call(func() C.int {
		return C.polyhip_scoring_create((*C.int32_t)(unsafe.Pointer(&lut[0])), (*C.uint8_t)(unsafe.Pointer(&validA[0])),
			(*C.uint8_t)(unsafe.Pointer(&validB[0])), C.int64_t(gap), &s.h)
	})
	if err != nil {
		return nil, err
	}
	runtime.SetFinalizer(s, func(s *Scoring) { s.Close() })
	return s, nil
}

func (s *Scoring) Close() {
	if s.h != nil {
		C.polyhip_scoring_destroy(s.h)
		s.h = nil
	}
}

// AlignResult is one pair's SmithWaterman outcome; Err != 0 encodes "Symbol X not in alphabet".
type AlignResult struct {
	Score        int64
	AlignA       string
	AlignB       string
	Err          uint32
	EndA, EndB   uint32
}

// SWAlignBatch: every A against one shared B (offB == nil) or pairwise.  The strings come back PACKED
// (polyhip_sw_align_batch_packed: only their own bytes cross PCIe, 0.3 GB instead of 1.05 GB per million 150-bp reads);
// a batch whose strings outgrow the first guess is run once more with the size the library reports.
func (s *Scoring) SWAlignBatch(A []byte, offA []uint64, B []byte, offB []uint64, maxLenA int) ([]AlignResult, error) {
	n := len(offA) - 1
	var pOffB *C.uint64_t
	shared := C.uint64_t(len(B))
	if offB != nil {
		pOffB = (*C.uint64_t)(unsafe.Pointer(&offB[0]))
		shared = 0
	}
	score := make([]int64, n+1)
	endA, endB, errs := make([]uint32, n+1), make([]uint32, n+1), make([]uint32, n+1)
	off := make([]uint64, n+1)
	capacity := uint64(len(A)) + uint64(len(A))/4 + 65536
	var alnA, alnB []byte
	for attempt := 0; ; attempt++ {
		alnA, alnB = make([]byte, capacity+1), make([]byte, capacity+1)
		var status C.int
		err := call(func() C.int {
			status = C.polyhip_sw_align_batch_packed(s.h, (*C.uint8_t)(unsafe.Pointer(&A[0])), (*C.uint64_t)(unsafe.Pointer(&offA[0])),
				C.uint64_t(n), (*C.uint8_t)(unsafe.Pointer(&B[0])), pOffB, shared, (*C.int64_t)(unsafe.Pointer(&score[0])),
				(*C.uint32_t)(unsafe.Pointer(&endA[0])), (*C.uint32_t)(unsafe.Pointer(&endB[0])),
				(*C.uint32_t)(unsafe.Pointer(&errs[0])), (*C.uint8_t)(unsafe.Pointer(&alnA[0])),
				(*C.uint8_t)(unsafe.Pointer(&alnB[0])), (*C.uint64_t)(unsafe.Pointer(&off[0])), C.uint64_t(capacity))
			return status
		})
		if err != nil && attempt == 0 && status == C.POLYHIP_ERR_INVALID && off[n] > capacity {
			capacity = off[n] // the strings did not fit: alnOff[npairs] is what they need
			continue
		}
		if err != nil {
			return nil, err
		}
		break
	}
	_ = maxLenA
	res := make([]AlignResult, n)
	for p := 0; p < n; p++ {
		res[p] = AlignResult{Score: score[p], AlignA: string(alnA[off[p]:off[p+1]]), AlignB: string(alnB[off[p]:off[p+1]]), Err: errs[p],
			EndA: endA[p], EndB: endB[p]}
	}
	return res, nil
}

// NWAlignBatch: align.NeedlemanWunsch (align.go:100-166) for every pair; B shared (offB == nil) or pairwise.
// EndA / EndB stay 0 (a global alignment has no end cell to report).
func (s *Scoring) NWAlignBatch(A []byte, offA []uint64, B []byte, offB []uint64, maxLenA int) ([]AlignResult, error) {
	n := len(offA) - 1
	lenB := uint64(len(B))
	var pOffB *C.uint64_t
	shared := C.uint64_t(len(B))
	if offB != nil {
		pOffB = (*C.uint64_t)(unsafe.Pointer(&offB[0]))
		shared, lenB = 0, 0
		for i := 0; i < n; i++ {
			if d := offB[i+1] - offB[i]; d > lenB {
				lenB = d
			}
		}
	}
	stride := maxLenA + int(lenB)
	if stride == 0 {
		stride = 1
	}
	score := make([]int64, n)
	errs, alen := make([]uint32, n), make([]uint32, n)
	alnA, alnB := make([]byte, n*stride+1), make([]byte, n*stride+1)
	err := call(func() C.int {
		return C.polyhip_nw_align_batch(s.h, (*C.uint8_t)(unsafe.Pointer(&A[0])), (*C.uint64_t)(unsafe.Pointer(&offA[0])),
			C.uint64_t(n), (*C.uint8_t)(unsafe.Pointer(&B[0])), pOffB, shared, (*C.int64_t)(unsafe.Pointer(&score[0])),
			(*C.uint32_t)(unsafe.Pointer(&errs[0])), (*C.uint8_t)(unsafe.Pointer(&alnA[0])),
			(*C.uint8_t)(unsafe.Pointer(&alnB[0])), (*C.uint32_t)(unsafe.Pointer(&alen[0])), C.uint32_t(stride))
	})
	if err != nil {
		return nil, err
	}
	res := make([]AlignResult, n)
	for p := 0; p < n; p++ {
		hi := (p + 1) * stride
		lo := hi - int(alen[p])
		res[p] = AlignResult{Score: score[p], AlignA: string(alnA[lo:hi]), AlignB: string(alnB[lo:hi]), Err: errs[p]}
	}
	return res, nil
}

// SantaLuciaBatch / SantaLuciaScan / MarmurDotyBatch / LeastRotationBatch follow the same pattern:

func SantaLuciaBatch(seqs []byte, offs []uint64, conc, na, mg float64) (tm, dH, dS []float64, err error) {
	n := len(offs) - 1
	tm, dH, dS = make([]float64, n), make([]float64, n), make([]float64, n)
	err = call(func() C.int {
		return C.polyhip_santalucia_batch((*C.uint8_t)(unsafe.Pointer(&seqs[0])), (*C.uint64_t)(unsafe.Pointer(&offs[0])),
			C.uint64_t(n), C.double(conc), C.double(na), C.double(mg), (*C.double)(unsafe.Pointer(&tm[0])),
			(*C.double)(unsafe.Pointer(&dH[0])), (*C.double)(unsafe.Pointer(&dS[0])))
	})
	return
}

func SantaLuciaScan(genome []byte, minLen, maxLen int, conc, na, mg float64) (tm, dH, dS []float64, ld int, err error) {
	ld = len(genome) - minLen + 1
	if ld < 0 {
		ld = 0
	}
	m := (maxLen-minLen+1)*ld + 1
	tm, dH, dS = make([]float64, m), make([]float64, m), make([]float64, m)
	err = call(func() C.int {
		return C.polyhip_santalucia_scan((*C.uint8_t)(unsafe.Pointer(&genome[0])), C.uint64_t(len(genome)), C.uint32_t(minLen),
			C.uint32_t(maxLen), C.double(conc), C.double(na), C.double(mg), (*C.double)(unsafe.Pointer(&tm[0])),
			(*C.double)(unsafe.Pointer(&dH[0])), (*C.double)(unsafe.Pointer(&dS[0])))
	})
	return
}

// SantaLuciaScanFirst: for every start of genome the first length in [minLen, maxLen] whose Tm is not below targetTm
// (0 = none) and that Tm (NaN where none): the grow loop of primers/pcr (pcr.go:47-53) at every position, reduced on
// the device -- 10 bytes per start come back instead of 24 per window.
func SantaLuciaScanFirst(genome []byte, minLen, maxLen int, conc, na, mg, targetTm float64) (firstLen []uint16, firstTm []float64, err error) {
	n := len(genome) - minLen + 1
	if n < 0 {
		n = 0
	}
	firstLen, firstTm = make([]uint16, n+1), make([]float64, n+1)
	err = call(func() C.int {
		return C.polyhip_santalucia_scan_first((*C.uint8_t)(unsafe.Pointer(&genome[0])), C.uint64_t(len(genome)), C.uint32_t(minLen),
			C.uint32_t(maxLen), C.double(conc), C.double(na), C.double(mg), C.double(targetTm),
			(*C.uint16_t)(unsafe.Pointer(&firstLen[0])), (*C.double)(unsafe.Pointer(&firstTm[0])))
	})
	return firstLen[:n], firstTm[:n], err
}

func MarmurDotyBatch(seqs []byte, offs []uint64) ([]float64, error) {
	n := len(offs) - 1
	tm := make([]float64, n)
	err := call(func() C.int {
		return C.polyhip_marmurdoty_batch((*C.uint8_t)(unsafe.Pointer(&seqs[0])), (*C.uint64_t)(unsafe.Pointer(&offs[0])),
			C.uint64_t(n), (*C.double)(unsafe.Pointer(&tm[0])))
	})
	return tm, err
}

func LeastRotationBatch(seqs []byte, offs []uint64) (rot []uint64, rotated []byte, err error) {
	n := len(offs) - 1
	rot = make([]uint64, n)
	rotated = make([]byte, len(seqs))
	err = call(func() C.int {
		return C.polyhip_least_rotation_batch((*C.uint8_t)(unsafe.Pointer(&seqs[0])), (*C.uint64_t)(unsafe.Pointer(&offs[0])),
			C.uint64_t(n), (*C.uint64_t)(unsafe.Pointer(&rot[0])), (*C.uint8_t)(unsafe.Pointer(&rotated[0])))
	})
	return
}

// SeqhashBatch: 71-character seqhashes ("" on error) and per-sequence error codes, see polyhip_seqhash_batch.
func SeqhashBatch(seqs []byte, offs []uint64, seqType int, circular, doubleStranded bool) ([]string, []uint32, error) {
	n := len(offs) - 1
	out := make([]byte, n*72+1)
	codes := make([]uint32, n+1)
	b2i := func(b bool) C.int {
		if b {
			return 1
		}
		return 0
	}
	err := call(func() C.int {
		return C.polyhip_seqhash_batch((*C.uint8_t)(unsafe.Pointer(&seqs[0])), (*C.uint64_t)(unsafe.Pointer(&offs[0])),
			C.uint64_t(n), C.int(seqType), b2i(circular), b2i(doubleStranded), (*C.char)(unsafe.Pointer(&out[0])),
			(*C.uint32_t)(unsafe.Pointer(&codes[0])))
	})
	res := make([]string, n)
	for i := 0; i < n; i++ {
		if codes[i] == 0 {
			res[i] = string(out[i*72 : i*72+71])
		}
	}
	return res, codes[:n], err
}

// ---- read feeders: a FASTQ / FASTA file image -> the packed batch the kernels take (io/fastq, io/fasta) ----

// Packed is a batch in the library's layout: sequence i = Seqs[Offsets[i]:Offsets[i+1]]; RecStart[i] is the byte
// offset of record i's identifier / header line in the file image (identifiers are sliced lazily from there).
type Packed struct {
	Seqs     []byte
	Offsets  []uint64
	RecStart []uint64
	Code     int    // 0, or the parse error's code (see polyhip_fastq_pack / polyhip_fasta_pack)
	Line     uint64 // FASTQ: the line the reference's message names
}

// FastqPack parses a whole FASTQ image on the device (io/fastq (*Parser).ParseAll semantics: the records
// before the first bad one are returned together with the error code).
func FastqPack(file []byte) (Packed, error) {
	most := len(file)/7 + 1 // shortest record: "@\nA\n\nI\n"
	p := Packed{Seqs: make([]byte, len(file)+1), Offsets: make([]uint64, most+2), RecStart: make([]uint64, most+1)}
	var result [4]uint64
	var pf *C.uint8_t
	if len(file) > 0 {
		pf = (*C.uint8_t)(unsafe.Pointer(&file[0]))
	}
	err := call(func() C.int {
		return C.polyhip_fastq_pack(pf, C.uint64_t(len(file)), (*C.uint8_t)(unsafe.Pointer(&p.Seqs[0])),
			(*C.uint64_t)(unsafe.Pointer(&p.Offsets[0])), (*C.uint64_t)(unsafe.Pointer(&p.RecStart[0])), C.uint64_t(most),
			(*C.uint64_t)(unsafe.Pointer(&result[0])))
	})
	if err != nil {
		return Packed{}, err
	}
	n := int(result[0])
	p.Code, p.Line = int(result[1]), result[2]
	p.Seqs, p.Offsets, p.RecStart = p.Seqs[:result[3]], p.Offsets[:n+1], p.RecStart[:n]
	return p, nil
}

// FastaPack parses a whole FASTA image on the device (io/fasta (*Parser).ParseAll semantics).
func FastaPack(file []byte) (Packed, error) {
	most := len(file)/2 + 2 // ">\n" is the shortest header line
	p := Packed{Seqs: make([]byte, len(file)+1), Offsets: make([]uint64, most+2), RecStart: make([]uint64, most+2)}
	var result [4]uint64
	var pf *C.uint8_t
	if len(file) > 0 {
		pf = (*C.uint8_t)(unsafe.Pointer(&file[0]))
	}
	err := call(func() C.int {
		return C.polyhip_fasta_pack(pf, C.uint64_t(len(file)), (*C.uint8_t)(unsafe.Pointer(&p.Seqs[0])),
			(*C.uint64_t)(unsafe.Pointer(&p.Offsets[0])), (*C.uint64_t)(unsafe.Pointer(&p.RecStart[0])), C.uint64_t(most),
			(*C.uint64_t)(unsafe.Pointer(&result[0])))
	})
	if err != nil {
		return Packed{}, err
	}
	n := int(result[0])
	p.Code = int(result[1])
	p.Seqs, p.Offsets, p.RecStart = p.Seqs[:result[2]], p.Offsets[:n+1], p.RecStart[:n]
	return p, nil
}

// ---- R1: all-gather of per-rank sketches (RCCL over xGMI; one process per GPU) ----

// Comm wraps polyhip_comm.  Rank 0 calls CommUniqueID and ships the 128 bytes to the other ranks.
type Comm struct{ c *C.polyhip_comm }

func CommUniqueID() ([128]byte, error) {
	var id [128]byte
	err := call(func() C.int { return C.polyhip_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0]))) })
	return id, err
}

func CommInitRank(id [128]byte, rank, nranks int) (*Comm, error) {
	cm := &Comm{}
	err := call(func() C.int {
		return C.polyhip_comm_init_rank((*C.uint8_t)(unsafe.Pointer(&id[0])), C.int(rank), C.int(nranks), &cm.c)
	})
	return cm, err
}

func (cm *Comm) Close() { C.polyhip_comm_destroy(cm.c); cm.c = nil }

// AllGatherV: ragged all-gather in place, rank r owns bytes [offsets[r], offsets[r+1]) of dBuf (device pointer).
func (cm *Comm) AllGatherV(dBuf unsafe.Pointer, offsets []uint64, stream unsafe.Pointer) error {
	return call(func() C.int {
		return C.polyhip_allgatherv_dev(cm.c, dBuf, (*C.uint64_t)(unsafe.Pointer(&offsets[0])), C.polyhip_stream_t(stream))
	})
}

// MashIndexBuildPart: rank r of n builds its part of the index of the gathered sketches dY (device) into dWork.
func MashIndexBuildPart(dY unsafe.Pointer, ny uint64, sy uint32, part, nparts uint32, dWork unsafe.Pointer, workBytes uint64, stream unsafe.Pointer) error {
	return call(func() C.int {
		return C.polyhip_mash_index_build_part_dev((*C.uint32_t)(dY), C.uint64_t(ny), C.uint32_t(sy), C.uint32_t(part),
			C.uint32_t(nparts), dWork, C.size_t(workBytes), C.polyhip_stream_t(stream))
	})
}

// MashIndexAllGather exchanges the parts (two ragged all-gathers) and finishes the index; then MashSharedCountsReuse.
func (cm *Comm) MashIndexAllGather(ny uint64, sy uint32, dWork unsafe.Pointer, workBytes uint64, stream unsafe.Pointer) error {
	return call(func() C.int {
		return C.polyhip_mash_index_allgather_dev(cm.c, C.uint64_t(ny), C.uint32_t(sy), dWork, C.size_t(workBytes),
			C.polyhip_stream_t(stream))
	})
}

// MashSharedCountsWorkspaceBytes sizes dWork for the calls above and below.
func MashSharedCountsWorkspaceBytes(nx uint64, sx uint32, ny uint64, sy uint32) uint64 {
	return uint64(C.polyhip_mash_shared_counts_workspace_bytes(C.uint64_t(nx), C.uint32_t(sx), C.uint64_t(ny), C.uint32_t(sy)))
}

// MashSharedCountsReuse: this rank's row block dX (device) against the index in dWork -> dCounts (u16, row stride ld).
func MashSharedCountsReuse(dX unsafe.Pointer, nx uint64, sx uint32, dY unsafe.Pointer, ny uint64, sy uint32, dCounts unsafe.Pointer,
	ld uint64, dWork unsafe.Pointer, workBytes uint64, stream unsafe.Pointer) error {
	return call(func() C.int {
		return C.polyhip_mash_shared_counts_reuse_dev((*C.uint32_t)(dX), C.uint64_t(nx), C.uint32_t(sx), (*C.uint32_t)(dY),
			C.uint64_t(ny), C.uint32_t(sy), (*C.uint16_t)(dCounts), C.uint64_t(ld), dWork, C.size_t(workBytes),
			C.polyhip_stream_t(stream))
	})
}

// AllGatherSketches takes DEVICE pointers (the sketches never leave HBM between K1 and K2).
func (cm *Comm) AllGatherSketches(dLocal unsafe.Pointer, nLocal uint64, s uint32, dAll unsafe.Pointer, stream unsafe.Pointer) error {
	return call(func() C.int {
		return C.polyhip_allgather_sketches_dev(cm.c, (*C.uint32_t)(dLocal), C.uint64_t(nLocal), C.uint32_t(s),
			(*C.uint32_t)(dAll), C.polyhip_stream_t(stream))
	})
}
