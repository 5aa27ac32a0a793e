{-# LANGUAGE DataKinds            #-}
{-# LANGUAGE FlexibleContexts     #-}
{-# LANGUAGE GADTs                #-}
{-# LANGUAGE InstanceSigs         #-}
{-# LANGUAGE KindSignatures       #-}
{-# LANGUAGE LambdaCase           #-}
{-# LANGUAGE PolyKinds            #-}
{-# LANGUAGE RankNTypes           #-}
{-# LANGUAGE ScopedTypeVariables  #-}
{-# LANGUAGE TypeApplications     #-}
{-# LANGUAGE TypeFamilies         #-}
{-# LANGUAGE TypeOperators        #-}

-- | @instance Tensor HipT@: the MI355X backend behind the reference's OUTER boundary, @class Tensor@
-- (@src/TensorOps/Types.hs:52-109@) -- the primary integration.  Nothing in @TensorOps.{Types,TOp,Tensor}@, the
-- @Learn@ modules or the type-level plumbing changes; the apps pick their backend with a 'Proxy'
-- (@app/Dots.hs:141-146@, @app/MNIST.hs:154@), so switching is @Proxy \@HipT@.
--
-- Tensors of every rank are flat device buffers (no nesting); handles carry their run-time dims, so the only
-- type-level evidence used is what the class hands over ('Length's as small ints, 'SingI' where a shape has no
-- operand to come from).
--
-- Laziness.  Inside 'trainBatch' / 'withScope' the class methods only RECORD (the library returns deferred
-- handles and fuses the recorded graph when a value is demanded -- @csrc/lazy.cpp@), which composes with Haskell's
-- own call-by-need: a thunk that is never forced records nothing, a recorded op nobody demands never runs -- and a
-- deferred handle that is merely still reachable (not yet finalised) is never launched either: neither the end of a
-- scope nor 'syncDevice' demands anything.  'rnf' of a tensor is @to_force@ (enqueue), 'forceMany' forces a product
-- in one plan; use 'syncDevice' where the reference's apps time a step (@app/MNIST.hs:413-420@).
--
-- NOT type-checked in this repository's build image (no GHC); the C++ mirror @tensor-ops_amd/host/tensorops/tensor.hpp@
-- is the tested rendering of exactly these bindings.
module TensorOps.Backend.HipTensor
  ( HipT(..)
  , E(..)
  , syncDevice, withScope
  , fromBatch, batchSum, gmulBatchSum
  , trainBatch, trainBatchReplay, trainAllOnline, liftH
  , commInit, allReduceSum
  ) where

import           Control.DeepSeq
import           Control.Monad                  (void)
import           Control.Monad.Primitive
import           Data.Kind
import           Data.List                      (foldl')
import           Data.Singletons
import           Data.Singletons.Prelude        (Sing(..))
import           Data.Singletons.TypeLits
import           Data.Type.Combinator
import           Data.Type.Length               as TCL
import           Data.Type.Product              as TCP
import           Data.Type.Uniform
import           Data.Type.Vector               (Vec, VecT(..))
import           Foreign
import           Foreign.C.Types
import qualified Foreign.Concurrent             as FC
import           Statistics.Distribution
import           System.IO.Unsafe               (unsafePerformIO)
import           System.Random.MWC
import           TensorOps.HIP.Expr
import           TensorOps.HIP.FFI
import           TensorOps.Learn.NeuralNet.FeedForward (Network(..), networkGradient)
import           TensorOps.Types
import           Type.Class.Higher
import           Type.Class.Higher.Util
import           Type.Family.List
import qualified Data.Finite                    as DF
import qualified Data.Finite.Internal           as DF
import qualified Data.Vector.Storable           as VS

newtype HipT (ns :: [Nat]) = HipT { unT :: H }

instance NFData (HipT ns) where
    rnf (HipT h) = unsafePerformIO (forceH h)
instance NFData1 HipT

instance Show (HipT ns) where
    showsPrec p (HipT h) = showParen (p > 10) $ showString "HipT " . shows (shapeOf h)
instance Show1 HipT

-- | Wait for everything enqueued so far (the @deepseq@ + wall clock of the apps).
syncDevice :: IO ()
syncDevice = chk c_sync

lenInt :: Length as -> CInt
lenInt = \case
    LZ   -> 0
    LS l -> 1 + lenInt l

vecList :: Vec n a -> [a]
vecList = \case
    ØV        -> []
    I x :* xs -> x : vecList xs

symVec :: Vec n a -> Vec n E
symVec = go 0
  where
    go :: Int -> Vec m a -> Vec m E
    go _ ØV          = ØV
    go i (_ :* rest) = I (S (XVar i)) :* go (i + 1) rest

toD :: E -> CDouble
toD (C a) = realToFrac a
toD (S _) = error "tensorops_hip: symbolic scalar where a number is needed"

dimsOf :: forall (ns :: [Nat]). SingI ns => p ns -> [Integer]
dimsOf _ = fromSing (sing :: Sing ns)

ixList :: Prod DF.Finite ns -> [Int64]
ixList = \case
    Ø       -> []
    i :< is -> fromIntegral (DF.getFinite i) : ixList is

-- | Every index of a shape in row-major order, first dim slowest (@genBTensorA@, @BTensor.hs:503-511@).
allIndices :: Sing (ns :: [Nat]) -> [Prod DF.Finite ns]
allIndices = \case
    SNil         -> [Ø]
    n `SCons` ns -> [ DF.Finite i :< is | i <- [0 .. fromSing n - 1], is <- allIndices ns ]

-- | An index of the leading dims from run-time integers (the 'Length' fixes how many).
mkIx :: Length ms -> [Integer] -> Prod DF.Finite ms
mkIx LZ     _        = Ø
mkIx (LS l) (i : is) = DF.Finite i :< mkIx l is
mkIx (LS _) []       = error "mkIx: too few indices"

constant :: [Integer] -> Double -> H
constant ds c = unsafePerformIO $ do
    dt <- elemDType
    withDims ds $ \r pd -> new1 (c_fill dt r pd 0 (realToFrac c))

instance Tensor HipT where
    type ElemT HipT = E

    liftT
        :: forall o n. SingI o
        => (Vec n E -> E)
        -> Vec n (HipT o)
        -> HipT o
    liftT f xs = HipT $ case xs of
        ØV -> constant (fromSing (sing :: Sing o)) (realToFrac (toD (f ØV)))          -- `TT.konst` (Tensor.hs:49-54)
        _  -> case f (symVec xs) of
                C c -> constant (fromSing (sing :: Sing o)) c
                S x -> let hs = map unT (vecList xs)
                       in unsafePerformIO $ withForeignPtr (compileX (length hs) x) $ \pe ->
                            withHs hs $ \k ph -> new1 (c_lift pe k ph)

    -- C[m,n] = sum_o A[m,o1..oq] B[oq..o1,n] (src/Data/Nested.hs:465-472); the three Length witnesses cross as
    -- small ints, the dims live in the handles
    gmul lM lO lN (HipT a) (HipT b) = HipT $ unsafePerformIO $ with2 a b $ \pa pb ->
        new1 (c_gmul (lenInt lM) (lenInt lO) (lenInt lN) pa pb)

    sumT :: forall o. SingI o => [HipT o] -> HipT o
    sumT xs = HipT $ unsafePerformIO $
        withHs (map unT xs) $ \k ph -> withDims (fromSing (sing :: Sing o)) $ \r pd -> new1 (c_sum k ph r pd)

    scaleT a (HipT x) = HipT $ unsafePerformIO $ withForeignPtr x $ \px -> new1 (c_scale (toD a) px)

    transp (HipT x) = HipT $ unsafePerformIO $ withForeignPtr x $ \px -> new1 (c_transp px)      -- zero-copy view

    sumRows (HipT x) = HipT $ unsafePerformIO $ withForeignPtr x $ \px -> new1 (c_sum_rows px)

    -- A host traversal over zero-copy row views -> f (device ops) -> one `to_stack`.  Call-by-need does the rest: the
    -- hot path's only use, the gradient of `TO.sumRows`, is `mapRows (LS LZ) (\_ -> dtdz) x`
    -- (src/TensorOps/TOp.hs:155-158) -- `f` ignores its row, so no `row is` thunk is ever forced (no to_slice call), and
    -- every element of `rows` is the SAME handle, which the library records as one broadcast node (csrc/api.cpp,
    -- to_stack) that the planner can fold into the loss head.  Nothing here knows that `f` is constant.
    mapRows l f (HipT x) = HipT $
        let k        = fromIntegral (lenInt l) :: Int
            (ds, _)  = shapeOf x
            lead     = take k ds
            ixs      = sequence [ [0 .. d - 1] | d <- lead ]
            row is   = HipT $ unsafePerformIO $ withForeignPtr x $ \px ->
                         withArray is $ \pi' -> new1 (c_slice px (fromIntegral k) pi')
            rows     = [ unT (f (row is)) | is <- ixs ]
        in unsafePerformIO $ withDims (map fromIntegral lead) $ \r pd -> withHs rows $ \_ ph -> new1 (c_stack r pd ph)

    diag u (HipT x) = HipT $ unsafePerformIO $ withForeignPtr x $ \px ->
        new1 (c_diag (1 + lenInt (uniformLength u)) px)

    getDiag _ (HipT x) = HipT $ unsafePerformIO $ withForeignPtr x $ \px -> new1 (c_get_diag px)

    -- as BTensor (BTensor.hs:841): draw on the host with the caller's generator, upload once.  (A host that does not
    -- need mwc-random's exact stream can use the device generator, `to_rand`: 'genRandDevice'.)
    genRand d g = generateA (\_ -> realToFrac <$> genContVar d g)

    generateA
        :: forall f ns. (Applicative f, SingI ns)
        => (Prod DF.Finite ns -> f E)
        -> f (HipT ns)
    -- (`gradTOp` seeds with `generateA (\_ -> I 1)`, src/TensorOps/Types.hs:132: an upload of up to 64 equal numbers is
    --  a constant the planner knows the value of, like `to_fill` -- csrc/api.cpp, to_from_host)
    generateA f = up <$> traverse f (allIndices (sing :: Sing ns))
      where
        up es = HipT $ unsafePerformIO $
                  fromHost (fromSing (sing :: Sing ns)) 0 (VS.fromList (map (realToFrac . toD) es))

    ixRows
        :: forall f ms os ns. (Applicative f, SingI (ms ++ os))
        => Length ms
        -> Length os
        -> (Prod DF.Finite ms -> HipT ns -> f (HipT os))
        -> HipT (ms ++ ns)
        -> f (HipT (ms ++ os))
    ixRows lM _ f (HipT x) =
        let k        = fromIntegral (lenInt lM) :: Int
            (ds, _)  = shapeOf x
            lead     = take k ds
            ixs      = sequence [ [0 .. d - 1] | d <- lead ]
            row is   = HipT $ unsafePerformIO $ withForeignPtr x $ \px ->
                         withArray is $ \pi' -> new1 (c_slice px (fromIntegral k) pi')
            stack rs = HipT $ unsafePerformIO $ withDims (map fromIntegral lead) $ \r pd ->
                         withHs (map unT rs) $ \_ ph -> new1 (c_stack r pd ph)
        in stack <$> traverse (\is -> f (mkIx lM (map fromIntegral is)) (row is)) ixs

    HipT x ! i = C . realToFrac $ unsafePerformIO $ withForeignPtr x $ \px ->
        withArray (ixList i) $ \pi' -> alloca $ \o -> chk (c_index px pi' 0 o) >> peek o

-- | The device's counter-based generator instead of the host's (`normalDistr 0 0.5` = dist 1, a 0, b 0.5;
-- FeedForward.hs:206-207).
genRandDevice :: forall ns. SingI ns => Int -> Double -> Double -> Word64 -> HipT ns
genRandDevice dist a b seed = HipT $ unsafePerformIO $ do
    dt <- elemDType
    withDims (fromSing (sing :: Sing ns)) $ \r pd ->
      new1 (c_rand dt r pd 0 (fromIntegral dist) (realToFrac a) (realToFrac b) seed)

-- ---- the batching extension (SURVEY.md 8(d)) --------------------------------------------------------------------
-- A handle may carry B independent samples of one logical shape; unbatched operands (parameters) broadcast, and
-- the cotangent of an unbatched input is the sum of its per-sample cotangents.  With batched x and y,
-- `gradTOp` of the reference's own `Network` is the batched gradient: nothing else in the DSL changes.

-- | B samples of shape ns, sample-major.
fromBatch :: forall ns. SingI ns => [VS.Vector Double] -> HipT ns
fromBatch rows = HipT $ unsafePerformIO $
    fromHost (fromSing (sing :: Sing ns)) (fromIntegral (length rows)) (VS.concat rows)

batchSum :: HipT ns -> HipT ns
batchSum (HipT x) = HipT $ unsafePerformIO $ withForeignPtr x $ \px -> new1 (c_batch_sum px)

-- | `gmul` followed by the sum over samples, fused (dW = sum_b dz_b (x) x_b is ONE GEMM with K = B).
gmulBatchSum :: Length ms -> Length os -> Length ns -> HipT (ms ++ os) -> HipT (Reverse os ++ ns) -> HipT (ms ++ ns)
gmulBatchSum lM lO lN (HipT a) (HipT b) = HipT $ unsafePerformIO $ with2 a b $ \pa pb ->
    new1 (c_gmul_batch_sum (lenInt lM) (lenInt lO) (lenInt lN) pa pb)

-- | One `trainNetwork` step (FeedForward.hs:131-148) on a BATCH.  The reference's own `networkGradient` (:166-176) is
-- called unchanged on batched x, y: the DSL knows nothing of batches, so the cotangent of every (unbatched) parameter
-- comes back batched -- for a weight matrix the per-sample outer products of TOp.hs:86-88.  The one thing the host adds
-- is 'batchSum' on each of them before the reference's update `p - r*g` (:145-147); inside the scope the library folds
-- that sum into the recorded contraction (`to_batch_sum` of a recorded gmul IS `to_gmul_batch_sum`), so the per-sample
-- value never exists.  The new parameters are forced TOGETHER before the scope closes: the whole step is one plan --
-- three launches for the MNIST stack.  Run it in a bound thread (the scope belongs to the OS thread).
--
-- Shapes are phantom on 'HipT', so the parameter product is rebuilt from plain handles ('reshapeProd').
trainBatch
    :: TOp '[ '[o], '[o] ] '[ '[] ]         -- ^ loss
    -> Double                                -- ^ rate
    -> HipT '[i] -> HipT '[o]                -- ^ batched inputs / targets ('fromBatch')
    -> Network HipT i o
    -> IO (Network HipT i o)
trainBatch loss r x y net = withScope $ case net of
    N s o p -> do
      let gs  = networkGradient loss x y net prodHandles          -- per-parameter cotangents, still thunks
          ps' = zipWith step (prodHandles p) gs
          step ph gh = liftH 2 (\[p0, g0] -> p0 - realToFrac r * g0) [ph, unT (batchSum (HipT gh))]
      forceMany ps'
      return (N s o (reshapeProd ps' p))

-- | The replayed form of 'trainBatch' for a loop over FIXED buffers: the step is issued once while the library
-- captures it (its three launches, with the update landing in the parameter buffers through @to_copy_into_many@), and
-- the returned action replays exactly those launches (~8 us of host time per step instead of the DSL's own cost).
-- New data is written into the buffers of x and y (@to_upload@ / @to_copy_into@), the parameters live in the buffers of
-- the network that was passed in.  This is the one place where the shim is not pure: handles derived from the old
-- parameter values are brought up to date by the library before the first replay overwrites them.
-- (Without it every 'trainBatch' call is still planned only once: the library keeps the plan of a recorded graph it has
-- seen before -- @to_plan_cache_stats@.)
trainBatchReplay
    :: TOp '[ '[o], '[o] ] '[ '[] ]
    -> Double
    -> HipT '[i] -> HipT '[o]
    -> Network HipT i o
    -> IO (IO ())
trainBatchReplay loss r x y net = case net of
    N _ _ p -> do
      let dsts = prodHandles p
          gs   = networkGradient loss x y net prodHandles
          step ph gh = liftH 2 (\[p0, g0] -> p0 - realToFrac r * g0) [ph, unT (batchSum (HipT gh))]
      chk c_graph_begin
      withScope $
        withHs dsts $ \n pd -> withHs (zipWith step dsts gs) $ \_ ps -> chk (c_copy_into_many n pd ps)
      -- (the capture -- a hipGraph, its executable and the tensors it retained -- lives as long as the launch action:
      --  a finaliser on the handle releases it; 'Foreign.Concurrent' because to_graph_release is a @safe@ import)
      g <- alloca (\pg -> chk (c_graph_end pg) >> peek pg) >>= \pg -> FC.newForeignPtr pg (void (c_graph_release pg))
      return (withForeignPtr g (chk . c_graph_launch))

-- | The reference's training loop -- @foldl' (\\nt (i,o) -> trainNetwork loss rate i o nt)@ over samples
-- (@app/MNIST.hs:390-396@) -- over rows @order@ of a resident batched data set.  ONE sample's step is captured through
-- the unchanged DSL (gradTOp on a hidden batch of one, the update, the new parameters copied into the old buffers) and
-- handed to the library: if the launches it planned for that step are an ffLayer stack's it trains the whole stream in one
-- persistent launch (@to_graph_online_sgd@, 10 us per sample on 784-300-100-10), otherwise the capture is replayed once
-- per sample with the sample copied into the staging buffers.  Nothing here says what the network is made of.
trainAllOnline
    :: TOp '[ '[o], '[o] ] '[ '[] ]
    -> Double
    -> HipT '[i] -> HipT '[o]               -- ^ resident data: B samples each
    -> [Int64]                              -- ^ sample order
    -> Network HipT i o                     -- ^ parameters: updated IN PLACE
    -> IO ()
trainAllOnline loss r xs ys order net = do
    let sample t i = HipT $ unsafePerformIO $ withForeignPtr (unT t) $ \p -> new1 (c_batch_select' p i)
        xbuf = sample xs 0                   -- staging buffers (a hidden batch of one)
        ybuf = sample ys 0
    step  <- trainBatchReplay' loss r xbuf ybuf net
    done  <- withForeignPtr (fst step) $ \pg -> with2 (unT xbuf) (unT ybuf) $ \px py -> with2 (unT xs) (unT ys) $ \pX pY ->
               withArrayLen order $ \n po -> alloca $ \ph ->
                 chk (c_graph_online_sgd pg px py pX pY (fromIntegral n) po ph) >> peek ph
    if done /= 0 then return () else
      mapM_ (\i -> do with2 (unT xbuf) (unT (sample xs i)) $ \d sr -> chk (c_copy_into d sr)
                       with2 (unT ybuf) (unT (sample ys i)) $ \d sr -> chk (c_copy_into d sr)
                       snd step) order
  where
    -- 'trainBatchReplay' that also hands out the graph handle
    trainBatchReplay' l rate x y n = case n of
      N _ _ p -> do
        let dsts = prodHandles p
            gs   = networkGradient l x y n prodHandles
            stp ph gh = liftH 2 (\[p0, g0] -> p0 - realToFrac rate * g0) [ph, unT (batchSum (HipT gh))]
        chk c_graph_begin
        withScope $ withHs dsts $ \k pd -> withHs (zipWith stp dsts gs) $ \_ ps -> chk (c_copy_into_many k pd ps)
        g <- alloca (\pg -> chk (c_graph_end pg) >> peek pg) >>= \pg -> FC.newForeignPtr pg (void (c_graph_release pg))
        return (g, withForeignPtr g (chk . c_graph_launch))

-- | `liftT` on plain handles (no 'SingI': shapes come from the operands).
liftH :: Int -> ([E] -> E) -> [H] -> H
liftH n f hs = case reify n f of
    Right (_, pe) -> unsafePerformIO $ withForeignPtr pe $ \ppe -> withHs hs $ \k ph -> new1 (c_lift ppe k ph)
    Left _        -> error "liftH: constant closure"

prodHandles :: Prod HipT ns -> [H]
prodHandles = \case
    Ø            -> []
    HipT h :< hs -> h : prodHandles hs

-- | The handles, element by element, under the shapes of an existing product.
reshapeProd :: [H] -> Prod HipT ns -> Prod HipT ns
reshapeProd hs = \case
    Ø        -> Ø
    _ :< ps' -> case hs of
                  h : rest -> HipT h :< reshapeProd rest ps'
                  []       -> error "reshapeProd: too few handles"

-- ---- data parallelism (SURVEY.md 8(e)): one process per GPU ------------------------------------------------------
-- Rank 0 makes the 128-byte id, the host program ships it to the other ranks over whatever transport it has,
-- every rank calls 'commInit'; per step: local batched gradient, ONE all-reduce of the flat buffer, identical update.

commInit :: Int -> Int -> VS.Vector Word8 -> IO ()
commInit rank world ident = VS.unsafeWith ident $ \p -> chk (c_comm_init (fromIntegral rank) (fromIntegral world) p)

allReduceSum :: HipT ns -> IO ()
allReduceSum (HipT g) = withForeignPtr g (chk . c_comm_allreduce_sum)
