{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE EmptyDataDecls           #-}

-- | Raw bindings to @libtensorops_hip.so@ (@include/tensorops_hip.h@) and the handle plumbing the two
-- class instances share ("TensorOps.BLAS.HIP", "TensorOps.Backend.HipTensor").
--
-- NOT type-checked in the build image of this repository (it has no GHC); every function below is
-- exercised through the same C entry points by the C++ mirror in @tensor-ops_amd/host/@ and by the Python
-- harness, which is where the behaviour is tested.  Written against GHC 8.0.1 / lts-7.2 like the reference
-- (@stack.yaml:18@).
--
-- Conventions (see the header): every entry point returns a status, nonzero = error with the text in
-- @to_last_error@ -- turned into a Haskell 'error' here, the reference's own failure mode for
-- "impossible" cases (@src/TensorOps/Tensor.hs:302@).  Handles are immutable values behind ref-counted
-- pointers: a 'ForeignPtr' with @to_release@ as finaliser, exactly like any other pure value under GC
-- (SURVEY.md 8(b) "Ownership").
--
-- Import modes (audited by @tests/test_hs_ffi.py@ against the table in INTEGRATION.md): an @unsafe@ call holds its
-- capability and delays every other capability at the next GC synchronisation for as long as it runs, so @unsafe@ is
-- only for entry points that are bounded by microseconds on every path -- queries, handle bookkeeping, and ops that
-- at most enqueue one kernel.  Anything that can wait for the stream, copy to or from the host, compile with hiprtc
-- (a closure on first use, a row program when a recorded graph is planned), plan and launch a recorded graph
-- (the in-place entry points run what still reads the memory they overwrite), or talk to other ranks is @safe@.
module TensorOps.HIP.FFI where

import           Control.Exception      (bracket_)
import           Control.Monad
import           Data.Int
import           Data.Word
import           Foreign
import           Foreign.C.String
import           Foreign.C.Types
import           System.IO.Unsafe       (unsafePerformIO)
import qualified Data.Vector.Storable   as VS

data ToTensor                       -- ^ @struct to_tensor_s@
data ToExpr                         -- ^ @struct to_expr_s@
data ToGraph                        -- ^ @struct to_graph_s@
type H  = ForeignPtr ToTensor
type HX = ForeignPtr ToExpr

tO_F32, tO_F64 :: CInt
tO_F32 = 0
tO_F64 = 1

-- ---- runtime -------------------------------------------------------------------------------------------
foreign import ccall safe   "to_init"              c_init           :: CInt -> IO CInt
foreign import ccall safe   "to_shutdown"          c_shutdown       :: IO CInt
foreign import ccall unsafe "to_last_error"        c_last_error     :: IO CString
foreign import ccall unsafe "to_device_count"      c_device_count   :: Ptr CInt -> IO CInt
foreign import ccall safe   "to_sync"              c_sync           :: IO CInt
foreign import ccall unsafe "to_stats"             c_stats          :: Ptr Int64 -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall unsafe "to_set_default_dtype" c_set_default_dtype :: CInt -> IO CInt
foreign import ccall unsafe "to_default_dtype"     c_default_dtype  :: Ptr CInt -> IO CInt
-- ---- handles ---------------------------------------------------------------------------------------------
foreign import ccall unsafe "to_alloc"        c_alloc       :: CInt -> CInt -> Ptr Int64 -> Int64 -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_retain"       c_retain      :: Ptr ToTensor -> IO CInt
foreign import ccall unsafe "&to_release"     p_release     :: FunPtr (Ptr ToTensor -> IO ())
foreign import ccall unsafe "to_shape"        c_shape       :: Ptr ToTensor -> Ptr CInt -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall unsafe "to_dtype"        c_dtype       :: Ptr ToTensor -> Ptr CInt -> IO CInt
foreign import ccall safe   "to_upload"       c_upload      :: Ptr ToTensor -> Ptr () -> Int64 -> IO CInt
foreign import ccall safe   "to_download"     c_download    :: Ptr ToTensor -> Ptr () -> Int64 -> IO CInt
foreign import ccall safe   "to_from_host"    c_from_host   :: CInt -> CInt -> Ptr Int64 -> Int64 -> Ptr () -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_fill"         c_fill        :: CInt -> CInt -> Ptr Int64 -> Int64 -> CDouble -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_rand"         c_rand        :: CInt -> CInt -> Ptr Int64 -> Int64 -> CInt -> CDouble -> CDouble -> Word64 -> Ptr (Ptr ToTensor) -> IO CInt
-- ---- class Tensor (src/TensorOps/Types.hs:52-109) -------------------------------------------------------
foreign import ccall safe "to_gmul"              c_gmul           :: CInt -> CInt -> CInt -> Ptr ToTensor -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall safe   "to_lift"            c_lift           :: Ptr ToExpr -> CInt -> Ptr (Ptr ToTensor) -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall safe "to_sum"               c_sum            :: CInt -> Ptr (Ptr ToTensor) -> CInt -> Ptr Int64 -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall safe "to_scale"             c_scale          :: CDouble -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_transp"          c_transp         :: Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_sum_rows"        c_sum_rows       :: Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_slice"           c_slice          :: Ptr ToTensor -> CInt -> Ptr Int64 -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_stack"           c_stack          :: CInt -> Ptr Int64 -> Ptr (Ptr ToTensor) -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_diag"            c_diag           :: CInt -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_get_diag"        c_get_diag       :: Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall safe   "to_index"           c_index          :: Ptr ToTensor -> Ptr Int64 -> Int64 -> Ptr CDouble -> IO CInt
foreign import ccall safe   "to_arg_max"         c_arg_max        :: Ptr ToTensor -> Ptr Int64 -> IO CInt
foreign import ccall safe   "to_arg_min"         c_arg_min        :: Ptr ToTensor -> Ptr Int64 -> IO CInt
foreign import ccall safe   "to_one_hot"         c_one_hot        :: CInt -> Int64 -> CDouble -> CDouble -> Int64 -> Ptr Int64 -> Ptr (Ptr ToTensor) -> IO CInt
-- ---- class BLAS (src/TensorOps/BLAS.hs:90-173) ---------------------------------------------------------
foreign import ccall unsafe "to_blas_axpy"      c_axpy      :: CDouble -> Ptr ToTensor -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall safe   "to_blas_dot"       c_dot       :: Ptr ToTensor -> Ptr ToTensor -> Ptr CDouble -> IO CInt
foreign import ccall unsafe "to_blas_ger"       c_ger       :: Ptr ToTensor -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_blas_gemv"      c_gemv      :: CDouble -> Ptr ToTensor -> Ptr ToTensor -> CDouble -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_blas_gemm"      c_gemm      :: CDouble -> Ptr ToTensor -> Ptr ToTensor -> CDouble -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_blas_scale"     c_bscale    :: CDouble -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_blas_add"       c_badd      :: Ptr ToTensor -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_blas_index_row" c_index_row :: Int64 -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_blas_transp"    c_btransp   :: Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_blas_eye"       c_eye       :: CInt -> Int64 -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall safe   "to_blas_trace"     c_trace     :: Ptr ToTensor -> Ptr CDouble -> IO CInt
foreign import ccall unsafe "to_blas_diag"      c_bdiag     :: Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_blas_get_diag"  c_bget_diag :: Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall safe   "to_blas_sum"       c_bsum      :: Ptr ToTensor -> Ptr CDouble -> IO CInt
-- ---- closures ----------------------------------------------------------------------------------------------
foreign import ccall safe   "to_expr_compile"  c_expr_compile :: CInt -> CInt -> Ptr Int32 -> CInt -> Ptr CDouble -> Ptr (Ptr ToExpr) -> IO CInt
foreign import ccall unsafe "&to_expr_release" p_expr_release :: FunPtr (Ptr ToExpr -> IO ())
-- ---- batching extension ------------------------------------------------------------------------------------
foreign import ccall safe "to_batch_sum"        c_batch_sum      :: Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_batch_bcast"    c_batch_bcast    :: Ptr ToTensor -> Int64 -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_batch_select"   c_batch_select   :: Ptr ToTensor -> Int64 -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_batch_slice"    c_batch_slice    :: Ptr ToTensor -> Int64 -> Int64 -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall safe   "to_batch_gather"   c_batch_gather   :: Ptr ToTensor -> Int64 -> Ptr Int64 -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_gmul_batch_sum" c_gmul_batch_sum :: CInt -> CInt -> CInt -> Ptr ToTensor -> Ptr ToTensor -> Ptr (Ptr ToTensor) -> IO CInt
-- ---- fusion scope, forcing, graph replay -------------------------------------------------------------------
foreign import ccall unsafe "to_memo_begin"   c_memo_begin   :: IO CInt
foreign import ccall unsafe "to_memo_end"     c_memo_end     :: IO CInt      -- demands nothing: what is still deferred stays deferred
foreign import ccall safe   "to_force"        c_force        :: Ptr ToTensor -> IO CInt
foreign import ccall safe   "to_force_many"   c_force_many   :: CInt -> Ptr (Ptr ToTensor) -> IO CInt
foreign import ccall unsafe "to_set_lazy"     c_set_lazy     :: CInt -> Ptr CInt -> IO CInt
foreign import ccall unsafe "to_set_loss_head_match" c_set_loss_head_match :: CInt -> Ptr CInt -> IO CInt
foreign import ccall unsafe "to_graph_begin"  c_graph_begin  :: IO CInt
foreign import ccall safe   "to_graph_end"    c_graph_end    :: Ptr (Ptr ToGraph) -> IO CInt
foreign import ccall safe   "to_graph_launch" c_graph_launch :: Ptr ToGraph -> IO CInt
foreign import ccall safe   "to_graph_release" c_graph_release :: Ptr ToGraph -> IO CInt
foreign import ccall safe   "to_graph_online_sgd" c_graph_online_sgd :: Ptr ToGraph -> Ptr ToTensor -> Ptr ToTensor -> Ptr ToTensor -> Ptr ToTensor -> Int64 -> Ptr Int64 -> Ptr CInt -> IO CInt
foreign import ccall unsafe "to_batch_select"  c_batch_select' :: Ptr ToTensor -> Int64 -> Ptr (Ptr ToTensor) -> IO CInt
-- ---- in-place program-level calls ---------------------------------------------------------------------------
foreign import ccall safe   "to_sgd_step_inplace" c_sgd_step_inplace :: Ptr ToTensor -> Ptr ToTensor -> CDouble -> IO CInt
foreign import ccall safe   "to_copy_into"        c_copy_into        :: Ptr ToTensor -> Ptr ToTensor -> IO CInt
foreign import ccall safe   "to_copy_into_many"   c_copy_into_many   :: CInt -> Ptr (Ptr ToTensor) -> Ptr (Ptr ToTensor) -> IO CInt
-- ---- data-parallel exchange -----------------------------------------------------------------------------------
foreign import ccall safe "to_comm_unique_id"     c_comm_unique_id     :: Ptr Word8 -> IO CInt
foreign import ccall safe "to_comm_init"          c_comm_init          :: CInt -> CInt -> Ptr Word8 -> IO CInt
foreign import ccall safe "to_comm_allreduce_sum" c_comm_allreduce_sum :: Ptr ToTensor -> IO CInt
foreign import ccall safe "to_comm_shutdown"      c_comm_shutdown      :: IO CInt
foreign import ccall safe "to_p2p_create"         c_p2p_create         :: Int64 -> CInt -> CInt -> Ptr Word8 -> IO CInt
foreign import ccall safe "to_p2p_connect"        c_p2p_connect        :: CInt -> Ptr Word8 -> IO CInt
foreign import ccall safe   "to_p2p_allreduce_sum" c_p2p_allreduce_sum :: Ptr ToTensor -> IO CInt
foreign import ccall safe   "to_p2p_allreduce_sgd" c_p2p_allreduce_sgd :: Ptr ToTensor -> Ptr ToTensor -> CDouble -> CInt -> IO CInt

-- ---- status -> error -----------------------------------------------------------------------------------------
chk :: IO CInt -> IO ()
chk act = do
    s <- act
    when (s /= 0) $ do
      msg <- c_last_error >>= peekCString
      error ("tensorops_hip: " ++ msg)

-- | The library is initialised once, on device 0 unless @TENSOROPS_HIP_DEVICE@ says otherwise (one process
-- per GPU: a data-parallel rank sets it to its local rank before the first tensor operation).
{-# NOINLINE hipReady #-}
hipReady :: ()
hipReady = unsafePerformIO $ chk (c_init 0)

-- | Run an entry point that produces one fresh handle.
new1 :: (Ptr (Ptr ToTensor) -> IO CInt) -> IO H
new1 f = hipReady `seq` alloca (\out -> chk (f out) >> peek out >>= newForeignPtr p_release)

-- | A pure class method: @unsafePerformIO@ is sound because every such entry point is a function of its
-- arguments (values are immutable, the library is stream-ordered and locks internally).
pure1 :: (Ptr (Ptr ToTensor) -> IO CInt) -> H
pure1 = unsafePerformIO . new1
{-# NOINLINE pure1 #-}

with2 :: H -> H -> (Ptr ToTensor -> Ptr ToTensor -> IO a) -> IO a
with2 a b f = withForeignPtr a $ \pa -> withForeignPtr b $ \pb -> f pa pb

-- | An array of handles for the n-ary entry points.
withHs :: [H] -> (CInt -> Ptr (Ptr ToTensor) -> IO a) -> IO a
withHs hs f = go hs []
  where
    go []     acc = withArrayLen (reverse acc) $ \n p -> f (fromIntegral n) p
    go (x:xs) acc = withForeignPtr x $ \p -> go xs (p : acc)

withDims :: [Integer] -> (CInt -> Ptr Int64 -> IO a) -> IO a
withDims ds f = withArrayLen (map fromIntegral ds) $ \n p -> f (fromIntegral n) p

-- | Run-time shape of a handle: dims and the hidden batch (0 = shared by all samples).
shapeOf :: H -> ([Int64], Int64)
shapeOf h = unsafePerformIO $ withForeignPtr h $ \p ->
    alloca $ \pr -> allocaArray 8 $ \pd -> alloca $ \pb -> do
      chk (c_shape p pr pd pb)
      r <- peek pr
      ds <- peekArray (fromIntegral r) pd
      b <- peek pb
      return (ds, b)

dtypeOf :: H -> CInt
dtypeOf h = unsafePerformIO $ withForeignPtr h $ \p -> alloca $ \o -> chk (c_dtype p o) >> peek o

-- | @ElemT@ of the instance: fp32 unless the program selected the fp64 instance (@to_set_default_dtype@).
elemDType :: IO CInt
elemDType = hipReady `seq` alloca (\o -> chk (c_default_dtype o) >> peek o)

-- | Host data, logical row-major (sample-major when batched), as doubles; converted to the instance's dtype.
fromHost :: [Integer] -> Int64 -> VS.Vector Double -> IO H
fromHost dims batch xs = do
    dt <- elemDType
    withDims dims $ \r pd ->
      if dt == tO_F64
        then VS.unsafeWith xs $ \px -> new1 (c_from_host dt r pd batch (castPtr px))
        else VS.unsafeWith (VS.map realToFrac xs :: VS.Vector Float) $ \px ->
               new1 (c_from_host dt r pd batch (castPtr px))

toHost :: H -> IO (VS.Vector Double)
toHost h = do
    let (ds, b) = shapeOf h
        n = fromIntegral (product ds * max 1 b) :: Int
    withForeignPtr h $ \p ->
      if dtypeOf h == tO_F64
        then do
          fp <- mallocForeignPtrArray n :: IO (ForeignPtr Double)
          withForeignPtr fp $ \q -> chk (c_download p (castPtr q) (fromIntegral (8 * n)))
          return (VS.unsafeFromForeignPtr0 fp n)
        else do
          fp <- mallocForeignPtrArray n :: IO (ForeignPtr Float)
          withForeignPtr fp $ \q -> chk (c_download p (castPtr q) (fromIntegral (4 * n)))
          return (VS.map realToFrac (VS.unsafeFromForeignPtr0 fp n))

-- | @rnf@ of one value for a lazy host: make its storage exist (enqueue), do not wait.
forceH :: H -> IO ()
forceH h = withForeignPtr h (chk . c_force)

-- | @rnf@ of a product of values in ONE call: the library plans them together, so launches they share (a weight
-- gradient and its bias gradient, the pair of weight-gradient GEMMs of a step) are shared.
forceMany :: [H] -> IO ()
forceMany hs = withHs hs $ \n p -> chk (c_force_many n p)

-- | A fusion scope (CSE memo + deferred, fused execution) around an action of the calling OS thread.
-- Bound threads only: the scope belongs to the OS thread, so run it inside 'Control.Concurrent.runInBoundThread'
-- (or on the main thread) when the RTS is @-threaded@.
--
-- Closing the scope (and 'TensorOps.Backend.HipTensor.syncDevice') demand NOTHING: under GC every intermediate of a
-- step is still reachable until the finaliser of its 'ForeignPtr' has run, and launching those would re-run the step
-- unfused.  Force what the step produces ('forceMany') before the scope closes; a deferred handle that is asked for
-- later is still computed then, on its own.
withScope :: IO a -> IO a
withScope = bracket_ (chk c_memo_begin) (chk c_memo_end)
